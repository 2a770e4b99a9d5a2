"""Console table + log.txt writer, byte-compatible with the reference's add_gym/util/logger.py:
the key set is fixed by the first row (:67-83), the console table prints floats as %8.3g and ints as
str (:86-114), log.txt rows are `{:<25}` columns of str(val) terminated by a carriage return (:116-143),
and with torch.distributed initialised every logged scalar is replaced by its mean over ranks, ints cast
back to int (:160-184).  Same class and method names, so code written against the reference's logger runs."""
import atexit
import os
import time

import torch
import torch.distributed as dist


class Logger:
    class Entry:
        def __init__(self, val, quiet=False):
            self.val = val
            self.quiet = quiet

    @staticmethod
    def is_root():
        return (not dist.is_initialized()) or dist.get_rank() == 0

    @staticmethod
    def print(msg, end=None):
        if Logger.is_root():
            print(msg, end=end)

    def __init__(self):
        self.output_file = None
        self.log_headers = []
        self.log_current_row = {}
        self._dump_str_template = ""
        self._max_key_len = 0
        self._row_count = 0
        self._need_update = True
        self._data_buffer = None

    def reset(self):
        self._row_count = 0
        self.log_headers = []
        self.log_current_row = {}
        self._need_update = True
        self._data_buffer = None
        if self.output_file is not None:
            self.output_file.truncate(0)

    def configure_output_file(self, filename=None):
        self._row_count = 0
        self.log_headers = []
        self.log_current_row = {}
        output_path = filename or "output/log_%i.txt" % int(time.time())
        out_dir = os.path.dirname(output_path)
        if Logger.is_root():
            if out_dir and not os.path.exists(out_dir):
                os.makedirs(out_dir, exist_ok=True)
            # newline="": the row terminator is a bare "\r" on every platform, as the reference writes it
            self.output_file = open(output_path, "w", newline="")
            atexit.register(self.output_file.close)
            Logger.print("Logging data to " + self.output_file.name)

    def log(self, key, val, quiet=False, **kwargs):
        if self._row_count == 0 and key not in self.log_headers:
            self.log_headers.append(key)
            self._max_key_len = max(self._max_key_len, len(key))
        elif key not in self.log_headers:
            raise AssertionError("Trying to introduce a new key %s that you didn't include in the first iteration" % key)
        self.log_current_row[key] = Logger.Entry(val, quiet)
        self._need_update = True

    def get_num_keys(self):
        return len(self.log_headers)

    def has_key(self, key):
        return key in self.log_headers

    def get_current_val(self, key):
        e = self.log_current_row.get(key)
        return None if e is None else e.val

    @staticmethod
    def format_console_value(val):
        if isinstance(val, float):
            return "%8.3g" % val
        if isinstance(val, int):
            return str(val)
        return val

    def console_lines(self):
        """The lines print_log prints (util/logger.py:93-114)."""
        w = self._max_key_len
        fmt = "| %" + str(w) + "s | %15s |"
        lines = ["-" * (22 + w)]
        for key in self.log_headers:
            e = self.log_current_row[key]
            if not e.quiet:
                lines.append(fmt % (key, Logger.format_console_value(e.val)))
        lines.append("-" * (22 + w))
        return lines

    def print_log(self):
        if dist.is_initialized() and self._need_update:
            self._mp_aggregate()
        if Logger.is_root():
            for line in self.console_lines():
                print(line)

    def row_strings(self):
        """(header string or None, value string) of the current row, without the "\\r" terminators."""
        template = "{:<25}" * self.get_num_keys()
        vals = [self.log_current_row[k].val for k in self.log_headers]
        head = template.format(*self.log_headers) if self._row_count == 0 else None
        return head, template.format(*map(str, vals))

    def write_log(self):
        if dist.is_initialized() and self._need_update:
            self._mp_aggregate()
        if Logger.is_root() and self.output_file is not None:
            head, row = self.row_strings()
            if head is not None:
                self.output_file.write(head + "\r")
            self.output_file.write(row + "\r")
            self.output_file.flush()
        self._row_count += 1

    def _mp_aggregate(self):
        """Mean over ranks of every logged scalar in one f64 all-reduce; int-typed entries stay ints (util/logger.py:160-184)."""
        if self._data_buffer is None:
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() and dist.get_backend() == "nccl" else torch.device("cpu")
            self._data_buffer = torch.zeros(len(self.log_headers), dtype=torch.float64, device=dev)
        host = torch.tensor([float(self.log_current_row[k].val) for k in self.log_headers], dtype=torch.float64)
        self._data_buffer.copy_(host)
        dist.all_reduce(self._data_buffer, op=dist.ReduceOp.SUM)
        self._data_buffer /= dist.get_world_size()
        for key, v in zip(self.log_headers, self._data_buffer.tolist()):
            e = self.log_current_row[key]
            e.val = int(v) if isinstance(e.val, int) else v
        self._need_update = False
