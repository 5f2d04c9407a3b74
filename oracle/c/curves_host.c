/* Host build of the product's curve-key header (cdsegnet_amd/csrc/curves.h) so that the exact
 * bit arithmetic the HIP kernels run can be compared with the numpy oracle WITHOUT a GPU
 * (tests/test_host_curves.py).  TEST INFRASTRUCTURE ONLY; built by __graft_entry__.build()
 * into oracle/_build/libcurves_host.so with gcc. */
#include <stdint.h>
#include "../../cdsegnet_amd/csrc/curves.h"

void cdseg_host_encode(const int64_t* grid, const int64_t* batch, long n, int depth, int order_id, int64_t* code) {
  for (long i = 0; i < n; ++i) {
    uint64_t k = curve_key(order_id, (uint32_t)grid[3 * i], (uint32_t)grid[3 * i + 1], (uint32_t)grid[3 * i + 2], depth);
    if (batch) k |= ((uint64_t)batch[i]) << (3 * depth);
    code[i] = (int64_t)k;
  }
}
