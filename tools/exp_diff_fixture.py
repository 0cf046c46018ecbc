#!/usr/bin/env python3
"""Diagnostic: decode one .lz4 file with the default (in-order) and the polling expand kernels and report where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libarchive_amd as la
from libarchive_amd.lz4 import decode_image
from libarchive_amd import _native as N
path = sys.argv[1]
img = open(path, "rb").read()
ctx = la.GpuContext(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
a, rca, msga, plana = decode_image(ctx, img, options=0)
b, rcb, msgb, planb = decode_image(ctx, img, options=N.LA_LZ4_OPT_EXPAND_POLL)
print("in-order:", rca, msga, a.size, " polling:", rcb, msgb, b.size)
ol, doff, bst, fst = plana.arrays()
print("block status (in-order) nonzero:", np.nonzero(bst)[0][:10], bst[np.nonzero(bst)[0][:10]])
n = min(a.size, b.size)
d = np.nonzero(a[:n] != b[:n])[0]
print("differing bytes:", d.size)
if d.size:
    first = int(d[0])
    blk = int(np.searchsorted(doff, first, side="right") - 1)
    print("first diff at", first, "block", blk, "offset in block", first - int(doff[blk]), "block out_len", int(ol[blk]))
    print("runs of diffs (first 10):")
    brk = np.nonzero(np.diff(d) > 1)[0]
    starts = np.concatenate(([0], brk + 1))[:10]
    ends = np.concatenate((brk, [d.size - 1]))[:10]
    for s_, e_ in zip(starts, ends):
        p = int(d[s_]); q = int(d[e_])
        bk = int(np.searchsorted(doff, p, side="right") - 1)
        print("  [%d, %d] len %d block %d off %d  got %s want %s" % (p, q, q - p + 1, bk, p - int(doff[bk]), a[p:p+8].tobytes().hex(), b[p:p+8].tobytes().hex()))

# map the first differing runs onto the block's sequences
def parse_block(p):
    i = 0; out = 0; seqs = []
    while i < len(p):
        tok = p[i]; i += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                c = p[i]; i += 1; ll += c
                if c != 255: break
        ls = i; i += ll
        d = out; out += ll
        if i >= len(p):
            seqs.append((d, ll, 0, 0, ls)); break
        off = p[i] | (p[i + 1] << 8); i += 2
        ml = (tok & 15) + 4
        if (tok & 15) == 15:
            while True:
                c = p[i]; i += 1; ml += c
                if c != 255: break
        seqs.append((d, ll, ml, off, ls)); out += ml
    return seqs
if d.size:
    idx = plana.index
    bk = int(np.searchsorted(doff, int(d[0]), side="right") - 1)
    blkrec = idx.blocks[bk]
    pay = img[int(blkrec["src_off"]):int(blkrec["src_off"]) + int(blkrec["src_len"])]
    seqs = parse_block(pay)
    print("block", bk, "sequences", len(seqs), "groups", (len(seqs) + 63) // 64)
    base = int(doff[bk])
    for s_, e_ in list(zip(starts, ends))[:8]:
        p = int(d[s_]) - base; q = int(d[e_]) - base
        for k, (dd, ll, ml, off, ls) in enumerate(seqs):
            if dd <= p < dd + ll + ml:
                part = "literal" if p < dd + ll else "match"
                print("  diff [%d,%d] in seq %d (group %d lane %d) %s: dst %d ll %d ml %d off %d lit_src %d" % (p, q, k, k // 64, k % 64, part, dd, ll, ml, off, ls))
                break
    # where do the wrong bytes come from?
    for s_, e_ in list(zip(starts, ends))[:6]:
        p = int(d[s_]); q = int(d[e_])
        if q - p + 1 < 3:
            continue
        gotb = a[p:q + 1].tobytes()
        pi = bytes(pay).find(gotb)
        oi = b.tobytes().find(gotb) if hasattr(b, "tobytes") else -1
        print("  run at %d len %d: got bytes found in this block's payload at %d, in the expected output at %d (this block starts at %d)" % (p - base, q - p + 1, pi, oi, base))
