"""Logging and checkpoints of the ADD agent, byte-compatible with the reference: log.txt rows and TensorBoard tags of
BaseAgent._log_train_info / PPOAgent._log_train_info / ADDAgent._log_train_info (base_agent.py:482-520, ppo_agent.py:277-279,
add/add_agent.py:235-265), checkpoint dicts of BaseAgent.save / load (base_agent.py:148-208) with the reference's tensor names, order, shapes
and dtypes (fixtures tests/golden/logger.npz, state_dict.npz).  Host-side only; mixed into learning.add_agent.ADDAgent."""
import os
import time

import numpy as np
import torch

from .. import _lib as L
from ..util.logger import Logger
from ..util.tb_logger import TBLogger


class AgentIO:
    def _build_logger(self, log_file):
        """base_agent.py:322-333: rank 0 logs to log.txt + TensorBoard (step key "Samples"); the other ranks keep a file-less
        Logger, which still takes part in the cross-rank mean of print_log / write_log."""
        if self._distributed and self._rank != 0:
            return Logger()
        log = TBLogger()
        log.set_step_key("Samples")
        log.configure_output_file(log_file)
        return log


    # ------------------------------------------------------------------ logging / checkpoints
    def _log_train_info(self, train_info, test_info, start_time):
        """base_agent.py:482-520 + ppo_agent.py:277-279: same keys, same order."""
        lg = self._logger
        ti = dict(train_info)
        lg.log("Iteration", int(self._iter), collection="1_Info")
        lg.log("Wall_Time", (time.time() - start_time) / 3600.0, collection="1_Info")
        lg.log("Samples", int(self._sample_count), collection="1_Info")
        lg.log("Test_Return", test_info["mean_return"], collection="0_Main")
        lg.log("Test_Episode_Length", test_info["mean_ep_len"], collection="0_Main", quiet=True)
        lg.log("Test_Episodes", int(test_info["num_eps"]), collection="1_Info", quiet=True)
        lg.log("Train_Return", ti.pop("mean_return"), collection="0_Main")
        lg.log("Train_Episode_Length", ti.pop("mean_ep_len"), collection="0_Main", quiet=True)
        lg.log("Train_Episodes", int(ti.pop("num_eps")), collection="1_Info", quiet=True)
        for k, v in ti.items():
            lg.log(k.title(), float(v))
        for k, v in self._env.get_diagnostics().items():  # base_agent.py:515-519
            lg.log(k.title(), float(v), collection="2_Env", quiet=True)
        lg.log("Exp_Prob", float(self._get_exp_prob()))  # ppo_agent.py:277-279
        if self._iter % self._iters_per_output == 0:  # add_agent.py:235-238
            self._log_sampler_distribution()

    def _log_sampler_distribution(self):
        """add_agent.py:240-265: bar charts of the per-segment mean error and mean start probability as a TensorBoard image
        (tag Sampler/Distribution, step = iteration).  Needs matplotlib; skipped silently without it, like any viewer extra."""
        lg = self._logger
        if not isinstance(lg, TBLogger) or lg._writer is None:
            return
        try:
            import io

            import matplotlib

            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            return
        err = self._smp["errors"]
        temp = self._smp_c.temperature
        tau = float(err.max()) + 1e-6 if temp <= 0 else temp  # sampler.py:57-73 over all clips
        probs = torch.softmax(err / tau, dim=-1)
        x = np.arange(self._num_segments)
        fig, (ax1, ax2) = plt.subplots(1, 2, figsize=(10, 3))
        ax1.bar(x, err.mean(dim=0).cpu().numpy())
        ax1.set_title("Mean Error per Segment")
        ax1.set_xlabel("Segment")
        ax2.bar(x, probs.mean(dim=0).cpu().numpy())
        ax2.set_title("Mean Prob per Segment")
        ax2.set_xlabel("Segment")
        fig.tight_layout()
        buf = io.BytesIO()
        fig.savefig(buf, format="png", dpi=100)
        plt.close(fig)
        w, h = fig.get_size_inches() * 100
        lg.add_image_png("Sampler/Distribution", int(h), int(w), buf.getvalue(), self._iter)

    def state_dict(self):
        Nm, tk = self._Nrm, self._task
        sd = {"_obs_norm._count": Nm["obs_cnt"].cpu(), "_obs_norm._mean": Nm["obs_mean"][:tk.obs_dim].cpu(), "_obs_norm._std": Nm["obs_std"][:tk.obs_dim].cpu(),
              "_a_norm._count": Nm["a_cnt"].cpu(), "_a_norm._mean": Nm["a_mean"][:L.NUM_DOF].cpu(), "_a_norm._std": Nm["a_std"][:L.NUM_DOF].cpu(),
              "_disc_obs_norm._count": Nm["d_cnt"].cpu(), "_disc_obs_norm._mean_abs": Nm["d_abs"][:tk.disc_dim].cpu()}
        sd.update(self._model.export())
        return sd

    def _optimizer_state_dict(self):
        """torch.optim.AdamW.state_dict() layout over the 22 trainable tensors in registration order (mp_optimizer.py:48-52)."""
        m = self._model
        ea, es = m.export(m.exp_avg), m.export(m.exp_avg_sq)
        keys = [k for k in ea if k != "_model._action_dist._logstd_net" or m.std_type == "CONSTANT"]  # (a FIXED log-std is not trainable: no optimiser state)
        if self._opt_type == "SGD":  # torch.optim.SGD.state_dict(): one momentum buffer per parameter
            state = {i: {"momentum_buffer": ea[k]} for i, k in enumerate(keys)} if m.opt_step > 0 else {}
            group = dict(lr=self._lr, momentum=0.9, dampening=0, weight_decay=self._wd, nesterov=False, maximize=False, foreach=None,
                         differentiable=False, fused=None, params=list(range(len(keys))))
            return {"state": state, "param_groups": [group]}
        state = {i: {"step": torch.tensor(float(m.opt_step)), "exp_avg": ea[k], "exp_avg_sq": es[k]} for i, k in enumerate(keys)} if m.opt_step > 0 else {}
        group = dict(lr=self._lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=self._wd, amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, decoupled_weight_decay=True, params=list(range(len(keys))))
        return {"state": state, "param_groups": [group]}

    def save(self, out_file):
        """base_agent.py:148-155: same dict layout and tensor names, so reference tooling (publish/push_to_hf.py) keeps working."""
        torch.save({"model": self.state_dict(), "optimizer": self._optimizer_state_dict(), "iter": self._iter, "sample_count": self._sample_count}, out_file)

    def load(self, in_file):
        """base_agent.py:157-208: accepts a full checkpoint or a bare state dict, with or without the DDP `.module` prefix."""
        ck = torch.load(in_file, map_location="cpu", weights_only=True)
        self._is_restored = True
        if "model" in ck and "optimizer" in ck:
            sd, opt = ck["model"], ck["optimizer"]
            self._iter, self._sample_count = int(ck.get("iter", 0)), int(ck.get("sample_count", 0))
        else:
            sd, opt, self._is_restored = ck, None, False
        sd = {k.replace("_model.module.", "_model."): v for k, v in sd.items()}
        Nm, tk, m = self._Nrm, self._task, self._model
        m.load(sd)
        m.refresh_shadow()
        Nm["obs_cnt"].copy_(sd["_obs_norm._count"])
        Nm["obs_mean"][:tk.obs_dim] = sd["_obs_norm._mean"].to(self._device)
        Nm["obs_std"][:tk.obs_dim] = sd["_obs_norm._std"].to(self._device)
        self._obs_norm_first = True  # mean_sq is rebuilt lazily like normalizer.py:38-39
        Nm["d_cnt"].copy_(sd["_disc_obs_norm._count"])
        Nm["d_abs"][:tk.disc_dim] = sd["_disc_obs_norm._mean_abs"].to(self._device)
        if opt is not None and opt.get("state") and "momentum_buffer" in opt["state"][0]:
            keys = [k for k in m.export() if k != "_model._action_dist._logstd_net" or m.std_type == "CONSTANT"]  # (a FIXED log-std is not trainable: no optimiser state)
            m.load({k: opt["state"][i]["momentum_buffer"] for i, k in enumerate(keys)}, m.exp_avg)
            m.opt_step = max(m.opt_step, 1)  # (torch's SGD keeps no step count: any value > 0 means "buffers are live")
        elif opt is not None and opt.get("state"):
            keys = [k for k in m.export() if k != "_model._action_dist._logstd_net" or m.std_type == "CONSTANT"]  # (a FIXED log-std is not trainable: no optimiser state)
            ea = {k: opt["state"][i]["exp_avg"] for i, k in enumerate(keys)}
            es = {k: opt["state"][i]["exp_avg_sq"] for i, k in enumerate(keys)}
            m.load(ea, m.exp_avg)
            m.load(es, m.exp_avg_sq)
            m.opt_step = int(float(opt["state"][0]["step"]))
        Logger.print(f"Loaded model parameters from {in_file}")

    def _output_train_model(self, it, out_model_file, int_output_dir):
        if self._rank != 0:
            return
        self.save(out_model_file)
        if int_output_dir != "":
            self.save(os.path.join(int_output_dir, "model_{:010d}.pt".format(it)))
