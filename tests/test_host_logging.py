"""Host-side compatibility surface, checked against fixtures generated from the imported reference
(tools/gen_golden_agent.py: gen_logger, gen_state_dict): log.txt bytes, console table, TensorBoard tags, checkpoint names/shapes."""
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import add_gym_amd  # noqa: E402,F401
from add_gym_amd.util import tb_logger  # noqa: E402
from tests.util import gload  # noqa: E402


def _js(g, k):
    return json.loads(bytes(g[k]).decode())


def test_log_txt_console_and_tags_are_byte_identical_to_the_reference(tmp_path, capsys):
    g = gload("logger")
    rows = _js(g, "rows")
    lg = tb_logger.TBLogger()
    lg.set_step_key("Samples")
    path = tmp_path / "log.txt"
    lg.configure_output_file(str(path))
    capsys.readouterr()
    consoles = []
    for row in rows:
        for key, val, col, quiet in row:
            lg.log(key, val, collection=col, quiet=quiet)
        lg.print_log()
        consoles.append(capsys.readouterr().out)
        lg.write_log()
    lg.output_file.flush()
    assert path.read_bytes() == bytes(g["log_txt"])        # util/logger.py:116-143: str(val), {:<25} columns, "\r" row ends
    assert consoles == _js(g, "console")                    # util/logger.py:86-114
    assert lg.key_tags() == _js(g, "tags")                  # util/tb_logger.py:63-74
    # the event file carries every key but the step key, under <collection>/<key>, at step = Samples
    ev = [f for f in os.listdir(tmp_path) if f.startswith("events.out.tfevents")]
    if isinstance(lg._writer, tb_logger.EventFileWriter):
        assert len(ev) == 1
        recs = tb_logger.read_events(os.path.join(tmp_path, ev[0]))
        tags = [t for t in _js(g, "tags") if not t.endswith("/Samples")]
        assert [r[1] for r in recs] == tags * len(rows)
        assert [r[0] for r in recs] == [131072] * len(tags) + [13238272] * len(tags)
        want = {t: float(np.float32(v)) for t, (k, v, c, q) in zip(_js(g, "tags"), rows[1])}
        for step, tag, val in recs[len(tags):]:
            assert val == want[tag], (tag, val, want[tag])


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert tb_logger.crc32c(b"123456789") == 0xE3069283
    assert tb_logger.crc32c(bytes(32)) == 0x8A9136AA
    assert tb_logger.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tb_logger.crc32c(bytes(range(32))) == 0x46DD794E


def test_int_entries_stay_ints_after_cross_rank_mean():
    """util/logger.py:178-183: the mean over ranks is cast back to int for int-typed entries (exercised without a process group by
    calling the per-entry rule directly)."""
    from add_gym_amd.util.logger import Logger

    lg = Logger()
    lg.log("Samples", 1048576)
    lg.log("Loss", 0.25)
    head, row = lg.row_strings()
    assert row.split() == ["1048576", "0.25"] and head.split() == ["Samples", "Loss"]


def test_checkpoint_names_shapes_dtypes_match_the_reference_agent():
    """The product's checkpoint layout against the reference agent's own state_dict / optimizer state_dict (fixture state_dict.npz)."""
    import torch
    from add_gym_amd.learning.model import Model
    from tests.test_dist_gloo import MODEL_CFG

    meta = _js(gload("state_dict"), "meta")
    m = Model(MODEL_CFG, 264, 272, 114, 128, torch.device("cpu"))
    exp = m.export()
    ref_model = {k: (tuple(shape), dt) for k, shape, dt in meta["model"]}
    norm_keys = {k for k in ref_model if not k.startswith("_model.")}
    assert norm_keys == {"_obs_norm._count", "_obs_norm._mean", "_obs_norm._std", "_a_norm._count", "_a_norm._mean", "_a_norm._std",
                         "_disc_obs_norm._count", "_disc_obs_norm._mean_abs"}
    assert set(exp) == set(ref_model) - norm_keys
    for k, v in exp.items():
        assert tuple(v.shape) == ref_model[k][0] and str(v.dtype) == ref_model[k][1], k
    # trainable tensors in the reference's registration order == the order of the optimizer state (mp_optimizer.py:48-52)
    keys = [k for k in exp if k != "_model._action_dist._logstd_net"]
    assert len(keys) == len(meta["opt_state"]) == 22
    for (i, st), k in zip(meta["opt_state"], keys):
        assert st["exp_avg"] == list(exp[k].shape), (i, k)
    assert meta["top"] == ["iter", "model", "optimizer", "sample_count"]
