"""Writes a minimal Nimble file ("tablet") for the reader tests: stripes of
streams, the Stripes / StripeGroup / Footer FlatBuffers of
dwio/nimble/tablet/Footer.fbs and the 20-byte postscript of
dwio/nimble/tablet/Postscript.cpp:67-85.  The reference's writer cannot be built
here, so this follows the schema files; FlatBuffers are emitted by the small
builder below (the flatbuffers package is not installed)."""
from __future__ import annotations

import struct

import numpy as np


class _Fb:
    """Forward FlatBuffers builder: objects are appended after the objects that
    point at them, so every uoffset is positive."""

    def __init__(self):
        self.b = bytearray(4)  # root uoffset, patched last

    def _align(self, a):
        while len(self.b) % a:
            self.b.append(0)

    def vector(self, values, fmt):
        size = struct.calcsize("<" + fmt)
        self._align(4)
        # the elements must be aligned to their size: pad so that pos + 4 is
        while (len(self.b) + 4) % size:
            self.b.extend(b"\0\0\0\0")
        pos = len(self.b)
        self.b += struct.pack("<I", len(values))
        for v in values:
            self.b += struct.pack("<" + fmt, v)
        return pos

    def string(self, s: bytes):
        self._align(4)
        pos = len(self.b)
        self.b += struct.pack("<I", len(s)) + s + b"\0"
        return pos

    def table(self, fields):
        """fields: list of None | ("scalar", fmt, value) | ("ref", builder_fn) in schema order.
        builder_fn() appends the child and returns its position (called after the table is laid out)."""
        self._align(8)
        # vtable
        slots, body = [], bytearray(4)  # soffset to vtable
        for f in fields:
            if f is None:
                slots.append(0)
                continue
            fmt = f[1] if f[0] == "scalar" else "I"
            size = struct.calcsize("<" + fmt)
            while len(body) % size:
                body.append(0)
            slots.append(len(body))
            body += struct.pack("<" + fmt, f[2] if f[0] == "scalar" else 0)
        vt = struct.pack("<HH", 4 + 2 * len(slots), len(body)) + b"".join(struct.pack("<H", s) for s in slots)
        self._align(2)
        vt_pos = len(self.b)
        self.b += vt
        self._align(8)  # the table 8-aligned so its uint64 fields are; padding after the vtable is fine
        pos = len(self.b)
        body[0:4] = struct.pack("<i", pos - vt_pos)
        self.b += body
        for f, slot in zip(fields, slots):
            if f is not None and f[0] == "ref":
                child = f[1]()
                at = pos + slot
                self.b[at:at + 4] = struct.pack("<I", child - at)
        return pos

    def vector_of_refs(self, builders):
        self._align(4)
        pos = len(self.b)
        self.b += struct.pack("<I", len(builders)) + bytes(4 * len(builders))
        for i, fn in enumerate(builders):
            child = fn()
            at = pos + 4 + 4 * i
            self.b[at:at + 4] = struct.pack("<I", child - at)
        return pos

    def finish(self, root):
        self.b[0:4] = struct.pack("<I", root)
        return bytes(self.b)


def _section(fb, offset, size):  # table MetadataSection {offset:uint64; size:uint32; compression_type; uncompressed_size}
    return fb.table([("scalar", "Q", offset), ("scalar", "I", size), ("scalar", "B", 0), ("scalar", "I", size)])


def write_tablet(stripes, stripes_per_group=2, optional=None):
    """stripes: list of (row_count, [stream bytes or b"" for an absent stream]).
    Returns the file bytes."""
    out = bytearray()
    offsets, sizes, rows, group_of = [], [], [], []
    stream_tables = []  # per stripe: (offsets, sizes)
    for i, (n_rows, streams) in enumerate(stripes):
        start = len(out)
        so, ss = [], []
        for s in streams:
            so.append(len(out) - start)
            ss.append(len(s))
            out += s
        offsets.append(start)
        sizes.append(len(out) - start)
        rows.append(n_rows)
        group_of.append(i // stripes_per_group)
        stream_tables.append((so, ss))
    n_streams = max((len(s) for _, s in stripes), default=0)
    group_sections = []
    for g in range((len(stripes) + stripes_per_group - 1) // stripes_per_group):
        members = [i for i in range(len(stripes)) if group_of[i] == g]
        so, ss = [], []
        for i in members:
            o, s = stream_tables[i]
            so += o + [0] * (n_streams - len(o))
            ss += s + [0] * (n_streams - len(s))
        fb = _Fb()
        root = fb.table([("scalar", "I", len(members)), ("ref", lambda so=so, fb=fb: fb.vector(so, "I")),
                         ("ref", lambda ss=ss, fb=fb: fb.vector(ss, "I"))])
        blob = fb.finish(root)
        group_sections.append((len(out), len(blob)))
        out += blob
    fb = _Fb()
    root = fb.table([("ref", lambda: fb.vector(rows, "I")), ("ref", lambda: fb.vector(offsets, "Q")),
                     ("ref", lambda: fb.vector(sizes, "I")), ("ref", lambda: fb.vector(group_of, "I"))])
    blob = fb.finish(root)
    stripes_section = (len(out), len(blob))
    out += blob
    opt_sections = []
    for name, data in (optional or {}).items():
        opt_sections.append((name, len(out), len(data)))
        out += data
    fb = _Fb()

    def optional_table():
        return fb.table([
            ("ref", lambda: fb.vector_of_refs([(lambda n=n: fb.string(n.encode())) for n, _, _ in opt_sections])),
            ("ref", lambda: fb.vector([o for _, o, _ in opt_sections], "Q")),
            ("ref", lambda: fb.vector([s for _, _, s in opt_sections], "I")),
            ("ref", lambda: fb.vector([0] * len(opt_sections), "B")),
            ("ref", lambda: fb.vector([s for _, _, s in opt_sections], "I")),
        ])

    root = fb.table([
        ("scalar", "Q", int(sum(rows))),
        ("ref", lambda: _section(fb, *stripes_section)),
        ("ref", lambda: fb.vector_of_refs([(lambda o=o, s=s: _section(fb, o, s)) for o, s in group_sections])),
        ("ref", optional_table) if opt_sections else None,
    ])
    footer = fb.finish(root)
    out += footer
    # postscript: u32 footer size | u8 footer compression | u8 checksum type | u64 checksum | u16 major | u16 minor | u16 magic
    out += struct.pack("<IBBQHHH", len(footer), 0, 0, 0, 0, 1, 0xA1FA)
    return bytes(out)
