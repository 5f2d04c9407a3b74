"""Oracle: integer side of the path (codes, orders, padding plan, pooling clusters).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy int64 throughout;
all results are bit-exact restatements of

* pointcept/models/utils/serialization/z_order.py:40-50, 66-101   (z-order key)
* pointcept/models/utils/serialization/hilbert.py:91-198          (Skilling Hilbert)
* pointcept/models/utils/serialization/default.py:8-24            (encode + batch bits)
* pointcept/models/utils/structure.py:47-102                      (Point.serialization)
* pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:188-244
                                                                  (get_padding_and_inverse)
* ...point_transformer_v3m1_base.py:464-505                       (pooling cluster structure)
* pointcept/models/utils/misc.py:11-28                            (offset <-> batch)
"""
import numpy as np

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


def offset2bincount(offset):
    offset = np.asarray(offset, dtype=np.int64)
    return np.diff(offset, prepend=0)


def offset2batch(offset):
    bc = offset2bincount(offset)
    return np.repeat(np.arange(len(bc), dtype=np.int64), bc)


def batch2offset(batch):
    return np.cumsum(np.bincount(np.asarray(batch, dtype=np.int64))).astype(np.int64)


def z_order_key(x, y, z, depth):
    """Bit interleave: x -> bit 3i+2, y -> 3i+1, z -> 3i, i < depth (z_order.py:40-50).

    The reference's 8-bit LUT form (z_order.py:66-101) drops coordinate bits at
    or above ``depth``; so does this loop.
    """
    x = np.asarray(x, dtype=np.int64)
    y = np.asarray(y, dtype=np.int64)
    z = np.asarray(z, dtype=np.int64)
    key = np.zeros_like(x)
    for i in range(depth):
        key |= ((x >> i) & 1) << (3 * i + 2)
        key |= ((y >> i) & 1) << (3 * i + 1)
        key |= ((z >> i) & 1) << (3 * i + 0)
    return key


def hilbert_key(grid, depth):
    """Skilling's transform over 3 x depth bit planes, then Gray->binary of the
    interleaved bit string (hilbert.py:143-198).  grid (N,3) int64 -> (N,) int64."""
    g = np.asarray(grid, dtype=np.int64)
    mask_all = (1 << depth) - 1
    X = [g[:, 0] & mask_all, g[:, 1] & mask_all, g[:, 2] & mask_all]
    X = [a.copy() for a in X]
    # bit index 0 in the reference is the MSB of the depth-bit word (hilbert.py:152-172)
    for bit in range(depth):
        q = 1 << (depth - 1 - bit)
        low = q - 1  # the "lower bits" slice gray[:, :, bit+1:]
        for dim in range(3):
            on = (X[dim] & q) != 0
            # where the bit is on: invert dim-0's lower bits
            X[0] = np.where(on, X[0] ^ low, X[0])
            # where it is off: exchange lower bits of dim 0 and dim `dim`
            t = np.where(on, 0, (X[0] ^ X[dim]) & low)
            X[dim] = X[dim] ^ t
            X[0] = X[0] ^ t
    # interleave: bit b (MSB first) of dims 0,1,2 -> consecutive positions (hilbert.py:175)
    h = np.zeros_like(X[0])
    for b in range(depth):  # b = significance inside the word, 0 = LSB
        for dim in range(3):
            h |= ((X[dim] >> b) & 1) << (3 * b + (2 - dim))
    # Gray -> binary = prefix XOR from the MSB (hilbert.py:68-88)
    shift = 1
    while shift < 3 * depth:
        h ^= h >> shift
        shift <<= 1
    return h


def encode(grid, batch, depth, order):
    """serialization/default.py:8-24."""
    g = np.asarray(grid, dtype=np.int64)
    if order == "z":
        code = z_order_key(g[:, 0], g[:, 1], g[:, 2], depth)
    elif order == "z-trans":
        code = z_order_key(g[:, 1], g[:, 0], g[:, 2], depth)
    elif order == "hilbert":
        code = hilbert_key(g, depth)
    elif order == "hilbert-trans":
        code = hilbert_key(g[:, [1, 0, 2]], depth)
    else:
        raise NotImplementedError(order)
    if batch is not None:
        code = (np.asarray(batch, dtype=np.int64) << (depth * 3)) | code
    return code


def serialization_depth(grid):
    """structure.py:66: int(grid_coord.max()).bit_length()."""
    return int(np.asarray(grid).max()).bit_length()


def serialization(grid, batch, orders=ORDERS, depth=None):
    """structure.py:47-93 *before* the shuffle: code, order, inverse, each (k, N)."""
    if depth is None:
        depth = serialization_depth(grid)
    nb = int(np.asarray(batch).max()) + 1 if len(batch) else 1
    assert depth * 3 + nb.bit_length() <= 63  # structure.py:69
    assert depth <= 16  # structure.py:74
    code = np.stack([encode(grid, batch, depth, o) for o in orders])
    order = np.argsort(code, axis=1, kind="stable")
    inverse = np.zeros_like(order)
    ar = np.arange(code.shape[1], dtype=np.int64)
    for k in range(code.shape[0]):
        inverse[k, order[k]] = ar
    return code, order.astype(np.int64), inverse.astype(np.int64), depth


def padding_plan(offset, patch_size):
    """ptv3.py:188-244 (flash-path semantics: fixed patch_size K).

    Returns pad (N',), unpad (N,), cu_seqlens (P+1,) as int64.
    Batch elements with more than K points are padded to a multiple of K; the
    missing slots of the last patch are filled with the *indices of the tail of
    the previous patch* (ptv3.py:218-228)."""
    offset = np.asarray(offset, dtype=np.int64)
    K = int(patch_size)
    bincount = offset2bincount(offset)
    bincount_pad = (bincount + K - 1) // K * K
    mask_pad = bincount > K
    bincount_pad = np.where(mask_pad, bincount_pad, bincount)
    _offset = np.concatenate([[0], offset])
    _offset_pad = np.concatenate([[0], np.cumsum(bincount_pad)])
    pad = np.arange(_offset_pad[-1], dtype=np.int64)
    unpad = np.arange(_offset[-1], dtype=np.int64)
    cu = []
    for i in range(len(offset)):
        unpad[_offset[i]:_offset[i + 1]] += _offset_pad[i] - _offset[i]
        if bincount[i] != bincount_pad[i]:
            r = int(bincount[i] % K)
            pad[_offset_pad[i + 1] - K + r:_offset_pad[i + 1]] = \
                pad[_offset_pad[i + 1] - 2 * K + r:_offset_pad[i + 1] - K]
        pad[_offset_pad[i]:_offset_pad[i + 1]] -= _offset_pad[i] - _offset[i]
        cu.append(np.arange(_offset_pad[i], _offset_pad[i + 1], K, dtype=np.int64))
    cu = np.concatenate(cu + [np.array([_offset_pad[-1]], dtype=np.int64)])
    return pad, unpad, cu


def pooling_structure(code, pooling_depth):
    """ptv3.py:477-492.  code (k,N) int64 (already shuffled by the caller).

    Returns cluster (N,), counts (M,), indices (N,) = a stable sort of cluster,
    idx_ptr (M+1,), head (M,), down_code (k,M), down_order, down_inverse.
    ``torch.sort(cluster)`` is unstable in the reference, so ``indices``/``head``
    are only defined up to the order inside a cluster; everything the forward
    reads through them is constant inside a cluster."""
    shifted = np.asarray(code, dtype=np.int64) >> (pooling_depth * 3)
    _, cluster, counts = np.unique(shifted[0], return_inverse=True, return_counts=True)
    cluster = cluster.astype(np.int64)
    indices = np.argsort(cluster, kind="stable").astype(np.int64)
    idx_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    head = indices[idx_ptr[:-1]]
    down_code = shifted[:, head]
    down_order = np.argsort(down_code, axis=1, kind="stable").astype(np.int64)
    down_inverse = np.zeros_like(down_order)
    ar = np.arange(down_code.shape[1], dtype=np.int64)
    for k in range(down_code.shape[0]):
        down_inverse[k, down_order[k]] = ar
    return cluster, counts.astype(np.int64), indices, idx_ptr, head, down_code, down_order, down_inverse
