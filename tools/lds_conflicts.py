"""Evaluate LDS bank-conflict degree of an access pattern on gfx950 (tables from
/opt/skills/guides/MI355X_MICROARCH.md §LDS).  addr_fn(lane) -> byte address."""
GROUPS_B128 = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]
GROUPS_32 = [list(range(32)), list(range(32, 64))]


def conflicts(addr_fn, width):
    """width in bytes: 4 (b32), 8 (b64), 16 (b128).  Returns total LDS cycles and ideal."""
    if width == 16:
        groups, nb = GROUPS_B128, 64
    elif width == 8:
        groups, nb = GROUPS_32, 64
    else:
        groups, nb = GROUPS_32, 32
    total = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addr_fn(l)
            for w in range(width // 4):
                b = ((a // 4) + w) % nb
                banks.setdefault(b, set()).add((a // 4) + w)
        total += max(len(s) for s in banks.values())
    return total, len(groups)


if __name__ == "__main__":
    # bf16 attention K tile: [key][16 bf16], lane (key=l&31, h=l>>5) reads 16 B
    for name, fn in {
        "K linear": lambda l: (l & 31) * 32 + (l >> 5) * 16,
        "K xor(key>>3)": lambda l: (l & 31) * 32 + (((l >> 5) ^ (((l & 31) >> 3) & 1)) * 16),
    }.items():
        print(name, conflicts(fn, 16))
    for stride in (2048, 2056, 2064, 2080):
        print("Vt stride", stride, conflicts(lambda l: min(l & 31, 17) * stride + (l >> 5) * 8, 8))
    # GEMM bf16 tile [row][32 bf16 + pad]: lane (row=l&15, chunk=l>>4) reads 16B
    for stride in (64, 80, 96, 144):
        print("gemm bf16 stride", stride, conflicts(lambda l: (l & 15) * stride + (l >> 4) * 16, 16))
    # GEMM f32 tile [row][32 f32 + pad]: lane (row=l&15, g=l>>4) reads 4B at k+g
    for stride in (128, 132, 136, 144):
        print("gemm f32 stride", stride, conflicts(lambda l: (l & 15) * stride + (l >> 4) * 4, 4))
    # fp32 attention: K [key][16 + pad f32] lane (key=l&15, g) reads 16B at g*16
    for stride in (64, 80, 96, 68 * 1):
        if stride % 16 == 0:
            print("attn32 K stride", stride, conflicts(lambda l: (l & 15) * stride + (l >> 4) * 16, 16))
    for stride in (4096, 4112, 4128, 4160):
        print("attn32 Vt stride", stride, conflicts(lambda l: (l & 15) * stride + (l >> 4) * 16, 16))
