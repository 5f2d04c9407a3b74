"""TEST INFRASTRUCTURE - CPU restatement of the device noise generator of the benchmark configuration
(`noise_source="device"`: csrc/elementwise.hip randn_kernel), so that the timed configuration's random draws are pinned
like everything else: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
reference implementation's known-answer vectors are checked in tests/test_oracle.py) followed by Box-Muller on 24-bit
uniforms.  The reference itself draws this tensor with torch.normal on the host (pointcept/models/default.py:393); the
device generator replaces the draw, not its distribution (tests/test_gpu_e2e.py checks moments and reproducibility).

Thread t (counter words t_lo, t_hi, offset_lo, offset_hi; key = seed_lo, seed_hi) produces four normals
out[4 t .. 4 t + 3] = (r0 cos, r0 sin, r1 cos, r1 sin) with r_h = sqrt(-2 ln u1), u1 = ((c[2h] >> 8) + 1) / 2^24 in
(0, 1], angle 2 pi (c[2h + 1] >> 8) / 2^24."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """counter (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32."""
    c = [np.asarray(counter[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k1
            n3 = p0.astype(np.uint32)
            c = [n0, n1, n2, n3]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def randn(n, seed, offset):
    """The n float32 normals cdseg_randn(out, n, seed, offset) writes."""
    nt = (n + 3) // 4
    t = np.arange(nt, dtype=np.uint64)
    ctr = np.stack([(t & np.uint64(0xFFFFFFFF)).astype(np.uint32), (t >> np.uint64(32)).astype(np.uint32),
                    np.full(nt, offset & 0xFFFFFFFF, dtype=np.uint32), np.full(nt, (offset >> 32) & 0xFFFFFFFF, dtype=np.uint32)], -1)
    key = np.stack([np.full(nt, seed & 0xFFFFFFFF, dtype=np.uint32), np.full(nt, (seed >> 32) & 0xFFFFFFFF, dtype=np.uint32)], -1)
    c = philox4x32_10(ctr, key)
    out = np.empty((nt, 4), dtype=np.float32)
    for h in range(2):
        u1 = ((c[:, 2 * h] >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
        u2 = (c[:, 2 * h + 1] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        r = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
        ang = np.float32(6.28318530717958647692) * u2
        out[:, 2 * h] = r * np.cos(ang, dtype=np.float32)
        out[:, 2 * h + 1] = r * np.sin(ang, dtype=np.float32)
    return out.reshape(-1)[:n]
