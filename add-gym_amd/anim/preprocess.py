"""Standalone motion preprocessor: clips (`.motion` files or a `motions.yaml` listing) -> the 100 Hz step tables the hot path
reads, written to the on-disk cache that every later launch (any rank) picks up instead of ingesting again.  CPU only; replaces
the reference's per-launch precompute and its `.pkl` side effect next to the dataset (`anim/motion_lib.py:164-320`).

    python -m add_gym_amd.anim.preprocess task.motion_file=<clip.motion|motions.yaml> task.motion_cache_dir=<dir> [task.reference_compat=false]

The tables are the ones `MotionLib` builds in-process (bit-identical to the reference's in compat mode: `tests/test_host_ingest.py`);
the cache key covers the clips' bytes, their weights, the joint order, dt and the kinematic tree.
"""
import sys
import time


def preprocess(cfg, verbose=True):
    from ..anim.kin_char_model import KinCharModel
    from ..anim.motion_lib import MotionLib

    task = cfg["task"]
    cache_dir = task.get("motion_cache_dir", None)
    if not cache_dir:
        raise ValueError("task.motion_cache_dir=<dir> is required: that is where the tables are written")
    kin = KinCharModel("cpu")
    kin.load_char_file(cfg["robot"]["urdf_path"])
    t0 = time.perf_counter()
    lib = MotionLib(task["motion_file"], list(task["motion_joint_order"]), kin, float(cfg["engine"]["ctrl_dt"]), "cpu",
                    reference_compat=task.get("reference_compat", True), cache_dir=cache_dir)
    dt = time.perf_counter() - t0
    if verbose:
        print(f"{lib.get_num_motions()} clip(s), {lib.total_steps} steps of {1.0 / lib._dt_inv:g} s = {2 * lib.total_steps * 36 * 4 / 1e6:.1f} MB of tables, "
              f"{'already cached' if lib.from_cache else 'ingested'} in {dt:.2f} s -> {cache_dir}")
    return lib


def main(argv=None):
    from ..config import load_config

    cfg = load_config("train", list(sys.argv[1:] if argv is None else argv))
    preprocess(cfg)


if __name__ == "__main__":
    main()
