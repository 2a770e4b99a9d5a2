"""Console table + log.txt writer behind the reference's `Logger` surface (add_gym/util/logger.py: class and method names,
argument meaning, and the bytes that reach the console and log.txt; pinned by tests/golden/logger.npz, which holds what the
reference's own Logger wrote for the same rows).

The format, as the fixture fixes it: the first row decides the columns and their order, a later row may not add one; the console
table is a dashed rule, one `| key | value |` line per non-quiet column (keys right-aligned to the longest key, values %8.3g for
floats and str() for ints, right-aligned to 15) and a closing rule; log.txt is 25-wide left-aligned columns of str(value), a
header line before the first row, every line ended by a bare carriage return.  Under torch.distributed every value is replaced by
its mean over ranks before it is shown, int-typed values staying ints, and only rank 0 prints or writes.

Own design below that surface: the row is a list of column records plus a name index; the cross-rank mean is one float64
all-reduce of the whole row.
"""
import atexit
import os
import time

import torch
import torch.distributed as dist

_COLUMN_WIDTH = 25  # log.txt
_VALUE_WIDTH = 15   # console


class _Column:
    __slots__ = ("name", "val", "quiet")

    def __init__(self, name):
        self.name, self.val, self.quiet = name, None, False


def _group_active():
    return dist.is_available() and dist.is_initialized()


class Logger:
    @staticmethod
    def is_root():
        return not _group_active() or dist.get_rank() == 0

    @staticmethod
    def print(msg, end=None):
        if Logger.is_root():
            print(msg, end=end)

    def __init__(self):
        self.output_file = None
        self._clear_rows()

    def _clear_rows(self):
        self._columns = []        # _Column records in first-row order
        self._where = {}          # name -> position in _columns
        self._rows_written = 0
        self._pending_mean = False  # values logged since the last cross-rank mean
        self._exchange = None

    # ---- configuration
    def reset(self):
        self._clear_rows()
        if self.output_file is not None:
            self.output_file.truncate(0)

    def configure_output_file(self, filename=None):
        self._clear_rows()
        path = filename if filename else "output/log_%i.txt" % int(time.time())
        if not Logger.is_root():
            return
        folder = os.path.dirname(path)
        if folder:
            os.makedirs(folder, exist_ok=True)
        self.output_file = open(path, "w", newline="")  # newline="": the "\r" line ends reach the file untranslated
        atexit.register(self.output_file.close)
        Logger.print("Logging data to " + self.output_file.name)

    # ---- one row
    def log(self, key, val, quiet=False, **kwargs):
        pos = self._where.get(key)
        if pos is None:
            assert self._rows_written == 0, "Trying to introduce a new key %s that you didn't include in the first iteration" % key
            pos = self._where[key] = len(self._columns)
            self._columns.append(_Column(key))
        col = self._columns[pos]
        col.val, col.quiet = val, quiet
        self._pending_mean = True

    def get_num_keys(self):
        return len(self._columns)

    def has_key(self, key):
        return key in self._where

    def get_current_val(self, key):
        pos = self._where.get(key)
        return None if pos is None else self._columns[pos].val

    def keys(self):
        return [c.name for c in self._columns]

    log_headers = property(keys)  # the reference's attribute name, read-only here
    rows_written = property(lambda self: self._rows_written)

    # ---- output
    @staticmethod
    def format_console_value(val):
        if isinstance(val, float):
            return "%8.3g" % val
        return str(val) if isinstance(val, int) else val

    def console_lines(self):
        """The lines print_log prints for the current row."""
        key_w = max((len(c.name) for c in self._columns), default=0)
        rule = "-" * (key_w + _VALUE_WIDTH + 7)
        body = ["| %*s | %*s |" % (key_w, c.name, _VALUE_WIDTH, Logger.format_console_value(c.val)) for c in self._columns if not c.quiet]
        return [rule] + body + [rule]

    def row_strings(self):
        """(header line or None, value line) of the current row, without their "\\r" terminators."""
        pad = lambda items: "".join(str(x).ljust(_COLUMN_WIDTH) for x in items)
        head = pad(c.name for c in self._columns) if self._rows_written == 0 else None
        return head, pad(c.val for c in self._columns)

    def print_log(self):
        self._mean_over_ranks()
        if Logger.is_root():
            print("\n".join(self.console_lines()))

    def write_log(self):
        self._mean_over_ranks()
        if Logger.is_root() and self.output_file is not None:
            head, row = self.row_strings()
            self.output_file.write((head + "\r" if head is not None else "") + row + "\r")
            self.output_file.flush()
        self._rows_written += 1

    def _mean_over_ranks(self):
        """Every logged scalar becomes its mean over ranks (one float64 all-reduce per row); int-typed entries stay ints."""
        if not (_group_active() and self._pending_mean):
            return
        if self._exchange is None or self._exchange.numel() != len(self._columns):
            on_gpu = torch.cuda.is_available() and dist.get_backend() == "nccl"
            self._exchange = torch.empty(len(self._columns), dtype=torch.float64,
                                         device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu"))
        self._exchange.copy_(torch.tensor([float(c.val) for c in self._columns], dtype=torch.float64))
        dist.all_reduce(self._exchange, op=dist.ReduceOp.SUM)
        world = dist.get_world_size()
        for c, total in zip(self._columns, self._exchange.tolist()):
            c.val = int(total / world) if isinstance(c.val, int) else total / world
        self._pending_mean = False
