#!/usr/bin/env python3
"""Lane-level numpy model of csrc/blockrr.hip's data mapping (v_mfma_f32_16x16x32_bf16 fragment layouts, weight
images, register-to-operand hand-offs), checked against plain matrix algebra.  Pure index logic: run on CPU.

MFMA D = A B, A (16 x 32), B (32 x 16): lane l = 16 q + i holds A[i][8q .. 8q+7] / B[8q .. 8q+7][i]; D[4q + r][i] in
register r of lane l."""
import numpy as np

rng = np.random.default_rng(0)


def mfma(afrag, bfrag, acc):
    """afrag, bfrag: (64, 8); acc (64, 4)."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        q, i = l >> 4, l & 15
        A[i, 8 * q:8 * q + 8] = afrag[l]
        B[8 * q:8 * q + 8, i] = bfrag[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        q, j = l >> 4, l & 15
        for r in range(4):
            out[l, r] += D[4 * q + r, j]
    return out


def p_chan(C, ct, i):  # "P layout": lane group q = i >> 2 owns C/4 consecutive channels
    return (i >> 2) * (C // 4) + ct * 4 + (i & 3)


def in_p(C, s, q, e):  # k-slot (q, e) of k step s when the operand is held in P layout
    return q * (C // 4) + 8 * s + e


def in_hid(u, q, e):  # k-slot (q, e) of hidden k step u when the operand comes from two natural 16-row tiles 2u, 2u+1
    return 32 * u + (4 * q + e if e < 4 else 16 + 4 * q + (e - 4))


def image(W, OT, KS, out_of, in_of):
    """fragment image: img[ot][s][lane] = 8 weights."""
    img = np.zeros((OT, KS, 64, 8))
    for ot in range(OT):
        for s in range(KS):
            for l in range(64):
                q, i = l >> 4, l & 15
                for e in range(8):
                    img[ot, s, l, e] = W[out_of(ot, i), in_of(s, q, e)]
    return img


def frags_from_p(T, C):
    """T: dict g -> acc array (CT, 64, 4) in P layout -> B fragments per k step: (KS, 64, 8)."""
    KS = C // 32
    out = np.zeros((KS, 64, 8))
    for s in range(KS):
        for l in range(64):
            out[s, l, :4] = T[2 * s, l]
            out[s, l, 4:] = T[2 * s + 1, l]
    return out


def run(C):
    n = 16  # one 16-point group
    KS, CT = C // 32, C // 16
    y = rng.normal(size=(n, C)); x = rng.normal(size=(n, C))
    Wl = rng.normal(size=(C, C)); Wq = rng.normal(size=(3 * C, C))
    W1 = rng.normal(size=(4 * C, C)); W2 = rng.normal(size=(C, 4 * C))
    # ---- head: t = y Wl^T in P layout
    yfrag = np.zeros((KS, 64, 8))
    for s in range(KS):
        for l in range(64):
            q, j = l >> 4, l & 15
            yfrag[s, l] = y[j, in_p(C, s, q, 0):in_p(C, s, q, 0) + 8]  # contiguous 16-byte load
    img_l = image(Wl, CT, KS, lambda ot, i: p_chan(C, ot, i), lambda s, q, e: in_p(C, s, q, e))
    t = np.zeros((CT, 64, 4))
    for ct in range(CT):
        for s in range(KS):
            t[ct] = mfma(img_l[ct, s], yfrag[s], t[ct])
    ref = y @ Wl.T
    for l in range(64):
        q, j = l >> 4, l & 15
        for ct in range(CT):
            for r in range(4):
                ch = q * (C // 4) + ct * 4 + r  # the lane's channels are consecutive in (ct, r) order
                assert abs(t[ct, l, r] - ref[j, ch]) < 1e-9
    # ---- h (P layout) -> qkv with lane-consecutive output channels
    h = t  # any tensor in P layout
    hfrag = frags_from_p(h, C)
    OT = 3 * C // 16
    img_q = image(Wq, OT, KS, lambda ot, i: (i >> 2) * (3 * C // 4) + ot * 4 + (i & 3), lambda s, q, e: in_p(C, s, q, e))
    refq = ref @ Wq.T
    for ot in range(OT):
        acc = np.zeros((64, 4))
        for s in range(KS):
            acc = mfma(img_q[ot, s], hfrag[s], acc)
        for l in range(64):
            q, j = l >> 4, l & 15
            for r in range(4):
                assert abs(acc[l, r] - refq[j, q * (3 * C // 4) + ot * 4 + r]) < 1e-8
    # ---- MLP: hidden tiles natural, fc2 consumes two tiles per k step straight from the accumulators
    img_1 = image(W1, 4 * C // 16, KS, lambda ht, i: 16 * ht + i, lambda s, q, e: in_p(C, s, q, e))
    img_2 = image(W2, CT, 4 * C // 32, lambda ot, i: p_chan(C, ot, i), lambda u, q, e: in_hid(u, q, e))
    acc2 = np.zeros((CT, 64, 4))
    hid_ref = np.maximum(ref @ W1.T, 0)  # stand-in non-linearity
    for u in range(4 * C // 32):
        a1 = np.zeros((2, 64, 4))
        for tt in range(2):
            for s in range(KS):
                a1[tt] = mfma(img_1[2 * u + tt, s], hfrag[s], a1[tt])
        a1 = np.maximum(a1, 0)
        Hf = np.concatenate([a1[0], a1[1]], 1)  # (64, 8): regs of tile 2u then tile 2u+1
        for ct in range(CT):
            acc2[ct] = mfma(img_2[ct, u], Hf, acc2[ct])
    ref2 = hid_ref @ W2.T
    for l in range(64):
        q, j = l >> 4, l & 15
        for ct in range(CT):
            for r in range(4):
                assert abs(acc2[ct, l, r] - ref2[j, q * (C // 4) + ct * 4 + r]) < 1e-7
    print("C =", C, "mapping OK")


if __name__ == "__main__":
    run(32)
    run(64)
