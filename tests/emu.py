"""Host-logic check helper: fills the device's output arrays with the ORACLE so that
the walker + stream-order resolution can be verified without a GPU (test-only)."""
import numpy as np

import oracle_lib as O
from libarchive_amd import _native as N


def emulate_lz4_device(image, idx):
    image = np.ascontiguousarray(np.frombuffer(bytes(image), dtype=np.uint8))
    nb, nf = len(idx.blocks), len(idx.frames)
    out_len = np.zeros(nb, np.uint32)
    bst = np.zeros(nb, np.uint32)
    fst = np.zeros(nf, np.uint32)
    outs = []
    prev = b""
    for i, b in enumerate(idx.blocks):
        pay = image[int(b["src_off"]):int(b["src_off"]) + int(b["src_len"])].tobytes()
        fl = int(b["flags"])
        if (fl & N.LA_LZ4B_CHECKSUM) and O.xxh32(pay) != int(b["block_sum"]):
            bst[i] = 1
            outs.append(b"")
            prev = b""
            continue
        if fl & N.LA_LZ4B_STORED:
            dec = pay
        else:
            d = b""
            if fl & N.LA_LZ4B_DEPENDENT:
                p = b"" if (fl & N.LA_LZ4B_FIRST) else prev[-65536:]
                d = bytes(65536 - len(p)) + p
            dec = O.lz4_block_decode(pay, int(b["dst_cap"]), d)
        if dec is None:
            bst[i] = 2
            dec = b""
        out_len[i] = len(dec)
        outs.append(dec)
        prev = dec
    dst_off = np.zeros(nb + 1, np.uint64)
    dst_off[1:] = np.cumsum(out_len.astype(np.uint64))
    for k, f in enumerate(idx.frames):
        if int(f["flags"]) & N.LA_LZ4F_HEADER_SUM:
            d = image[int(f["desc_off"]):int(f["desc_off"]) + int(f["desc_len"])].tobytes()
            if ((O.xxh32(d[:-1]) >> 8) & 0xFF) != d[-1]:
                fst[k] = 3
                continue
        if int(f["flags"]) & N.LA_LZ4F_CONTENT_SUM:
            a, n = int(f["first_block"]), int(f["n_blocks"])
            if O.xxh32(b"".join(outs[a:a + n])) != int(f["content_sum"]):
                fst[k] = 4
    return out_len, dst_off, bst, fst, b"".join(outs)
