"""TBLogger: the reference's add_gym/util/tb_logger.py surface (collections -> tags `<collection>/<key>`,
step = the value of the step key, the step key itself is not plotted; :30-74) on top of Logger.

The reference writes through torch.utils.tensorboard.SummaryWriter.  That package is optional here: when it
is not importable, scalars (and the sampler-distribution image) go through EventFileWriter below, a
dependency-free writer of the TensorBoard event-file format (TFRecord framing with masked CRC-32C,
hand-encoded Event / Summary protobuf messages), so dashboards pointed at the log directory keep working."""
import os
import socket
import struct
import time

from . import logger


# ---------------------------------------------------------------- CRC-32C (Castagnoli), table driven
def _crc_table():
    tab = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC = _crc_table()


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------- protobuf wire format (only what Event/Summary need)
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _bytes_field(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _varint_field(num, val):
    return _varint((num << 3) | 0) + _varint(val)


def scalar_value(tag, value):
    """Summary.Value{tag=1, simple_value=2}"""
    return _bytes_field(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))


def image_value(tag, height, width, png_bytes, colorspace=4):
    """Summary.Value{tag=1, image=4: Summary.Image{height=1, width=2, colorspace=3, encoded_image_string=4}}"""
    img = _varint_field(1, height) + _varint_field(2, width) + _varint_field(3, colorspace) + _bytes_field(4, png_bytes)
    return _bytes_field(1, tag.encode()) + _bytes_field(4, img)


def event(wall_time, step=None, summary_values=None, file_version=None):
    """Event{wall_time=1 (double), step=2 (int64), file_version=3 (string), summary=5 (Summary{value=1 repeated})}"""
    out = _varint((1 << 3) | 1) + struct.pack("<d", wall_time)
    if step is not None:
        out += _varint_field(2, int(step))
    if file_version is not None:
        out += _bytes_field(3, file_version.encode())
    if summary_values:
        out += _bytes_field(5, b"".join(_bytes_field(1, v) for v in summary_values))
    return out


class EventFileWriter:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        name = "events.out.tfevents.%010d.%s.%d.0" % (int(time.time()), socket.gethostname(), os.getpid())
        self.path = os.path.join(log_dir, name)
        self._f = open(self.path, "wb")
        self._record(event(time.time(), file_version="brain.Event:2"))

    def _record(self, data):
        head = struct.pack("<Q", len(data))
        self._f.write(head + struct.pack("<I", masked_crc(head)) + data + struct.pack("<I", masked_crc(data)))

    def add_scalar(self, tag, value, step):
        self._record(event(time.time(), step, [scalar_value(tag, value)]))

    def add_image_png(self, tag, height, width, png_bytes, step):
        self._record(event(time.time(), step, [image_value(tag, height, width, png_bytes)]))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    """[(step, tag, simple_value or None)] of an event file; verifies both CRCs of every record (test helper)."""
    def fields(buf):
        i = 0
        while i < len(buf):
            key, n = 0, 0
            while True:
                b = buf[i]
                i += 1
                key |= (b & 0x7F) << n
                n += 7
                if not b & 0x80:
                    break
            num, wt = key >> 3, key & 7
            if wt == 0:
                v, n = 0, 0
                while True:
                    b = buf[i]
                    i += 1
                    v |= (b & 0x7F) << n
                    n += 7
                    if not b & 0x80:
                        break
                yield num, v
            elif wt == 1:
                yield num, buf[i:i + 8]
                i += 8
            elif wt == 5:
                yield num, buf[i:i + 4]
                i += 4
            else:
                ln, n = 0, 0
                while True:
                    b = buf[i]
                    i += 1
                    ln |= (b & 0x7F) << n
                    n += 7
                    if not b & 0x80:
                        break
                yield num, buf[i:i + ln]
                i += ln

    out = []
    with open(path, "rb") as f:
        raw = f.read()
    i = 0
    while i < len(raw):
        head = raw[i:i + 8]
        (ln,) = struct.unpack("<Q", head)
        assert struct.unpack("<I", raw[i + 8:i + 12])[0] == masked_crc(head), "length CRC"
        data = raw[i + 12:i + 12 + ln]
        assert struct.unpack("<I", raw[i + 12 + ln:i + 16 + ln])[0] == masked_crc(data), "data CRC"
        i += 16 + ln
        ev = dict(fields(data))
        if 5 in ev:
            for num, val in fields(ev[5]):
                v = dict(fields(val))
                simple = struct.unpack("<f", v[2])[0] if 2 in v else None
                out.append((ev.get(2, 0), v[1].decode(), simple))
    return out


class TBLogger(logger.Logger):
    MISC_TAG = "Misc"

    def __init__(self):
        super().__init__()
        self._writer = None
        self._step_key = None
        self._collections = dict()

    def configure_output_file(self, filename=None):
        super().configure_output_file(filename)
        if logger.Logger.is_root():
            out_dir = os.path.dirname(filename) or "."
            try:
                from torch.utils.tensorboard import SummaryWriter  # the reference's writer, when installed

                self._writer = SummaryWriter(out_dir)
            except Exception:
                self._writer = EventFileWriter(out_dir)

    def set_step_key(self, var_key):
        self._step_key = var_key

    def log(self, key, val, collection=None, quiet=False):
        super().log(key, val, quiet)
        if collection is not None:
            self._collections.setdefault(collection, []).append(key)

    def key_tags(self):
        tags = []
        for key in self.keys():
            tag = TBLogger.MISC_TAG
            for col, keys in self._collections.items():
                if key in keys:
                    tag = col
            tags.append("{:s}/{:s}".format(tag, key))
        return tags

    def write_log(self):
        row_count = self.rows_written
        super().write_log()
        if logger.Logger.is_root() and self._writer is not None:
            if row_count == 0:
                self._key_tags = self.key_tags()
            step = row_count if self._step_key is None else self.get_current_val(self._step_key)
            for tag, key in zip(self._key_tags, self.keys()):
                if key != self._step_key:
                    self._writer.add_scalar(tag, self.get_current_val(key), step)
            self._writer.flush()

    def add_image_png(self, tag, height, width, png_bytes, step):
        """Sampler/Distribution bar charts (add_agent.py:240-265) arrive here as PNG bytes."""
        if self._writer is None:
            return
        if isinstance(self._writer, EventFileWriter):
            self._writer.add_image_png(tag, height, width, png_bytes, step)
        else:  # SummaryWriter wants a CHW float tensor
            import io

            import numpy as np
            from PIL import Image

            img = np.asarray(Image.open(io.BytesIO(png_bytes)).convert("RGB"), dtype=np.float32) / 255.0
            self._writer.add_image(tag, img.transpose(2, 0, 1), step)
