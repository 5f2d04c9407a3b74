"""CPU oracle for the CDSegNet single-step-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cdsegnet_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported baseline, never
as the thing measured or shipped.

It is a restatement (own code, numpy + PyTorch-CPU fp32) of the reference's
algorithm for this path; every function cites the reference file:line it
follows.  It is pinned against golden vectors captured by running the
reference's own Python in the build container (``oracle/make_golden.py`` ->
``tests/golden/*.npz``).

Pinning status
--------------
* serialization (z-order / Hilbert codes, orders, inverses), padding plan,
  pooling cluster structure, ``calc_t_emb``: pinned bit-exactly by the
  reference's own code (fixtures ``serialization_*.npz``, ``padding_*.npz``).
* attention core, LayerNorm/Linear/GELU blocks, pooling / unpooling, cross
  block, full forward: pinned against the reference's PyTorch-CPU fp32
  (non-flash) execution (fixtures ``mini_e2e_*.npz``, ``full_e2e_8k.npz``).
* sparse convolution (spconv.SubMConv3d): spconv is a third-party package that
  is neither vendored under /root/reference nor installed here and the
  reference holds no test for it -> **parity unpinned** for the conv weight
  layout / offset convention.  The oracle's convention (cross-correlation,
  weight ``(out, kx, ky, kz, in)``, kernel axis a <-> grid axis a) is checked
  against a dense ``torch.nn.functional.conv3d`` instead.
"""
