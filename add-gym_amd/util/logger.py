"""Console table + log.txt writer with the reference's row format (util/logger.py:67-143: keys fixed after
the first row, `{:<25}` columns, rows appended to the log file) and cross-rank mean of the logged scalars
(util/logger.py:160-184).  TensorBoard is optional (written only if the package is importable)."""
import os

import torch
import torch.distributed as dist


class Logger:
    @staticmethod
    def is_root():
        return (not dist.is_initialized()) or dist.get_rank() == 0

    @staticmethod
    def print(msg, end=None):
        if Logger.is_root():
            print(msg, end=end)

    def __init__(self, log_file=None, world=1):
        self._file = None
        self._headers, self._row, self._quiet = [], {}, set()
        self._row_count, self._world = 0, world
        self._collections = {}
        self._tb = None
        if log_file is not None and Logger.is_root():
            os.makedirs(os.path.dirname(str(log_file)) or ".", exist_ok=True)
            self._file = open(log_file, "w")
            try:
                from torch.utils.tensorboard import SummaryWriter  # optional dependency

                self._tb = SummaryWriter(os.path.dirname(str(log_file)))
            except Exception:
                self._tb = None

    def log(self, key, val, collection=None, quiet=False):
        if self._row_count == 0 and key not in self._headers:
            self._headers.append(key)
        elif key not in self._headers:
            raise KeyError(f"new log key {key!r} after the first row (the reference fixes the key set, util/logger.py:72-79)")
        self._row[key] = float(val)
        if quiet:
            self._quiet.add(key)
        if collection is not None:
            self._collections[key] = collection

    def _aggregate(self):
        if self._world > 1 and dist.is_initialized():
            keys = list(self._headers)
            t = torch.tensor([self._row.get(k, 0.0) for k in keys], dtype=torch.float64, device="cuda")
            dist.all_reduce(t)
            t /= self._world
            for k, v in zip(keys, t.tolist()):
                self._row[k] = v

    def print_log(self):
        self._aggregate()
        if not Logger.is_root():
            return
        keys = [k for k in self._headers if k not in self._quiet]
        width = max(len(k) for k in keys) if keys else 10
        line = "-" * (width + 22)
        print(line)
        for k in keys:
            print("| {:<{w}} | {:>15.6g} |".format(k, self._row.get(k, 0.0), w=width))
        print(line)

    def write_log(self):
        if self._file is None:
            self._row_count += 1
            return
        if self._row_count == 0:
            self._file.write("".join("{:<25}".format(k) for k in self._headers) + "\n")
        self._file.write("".join("{:<25}".format("{:.6g}".format(self._row.get(k, 0.0))) for k in self._headers) + "\n")
        self._file.flush()
        if self._tb is not None:
            step = int(self._row.get("Samples", self._row_count))
            for k in self._headers:
                self._tb.add_scalar("{}/{}".format(self._collections.get(k, "Misc"), k), self._row.get(k, 0.0), step)
        self._row_count += 1
