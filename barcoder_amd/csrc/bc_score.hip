// bc_score.hip — mismatch-efficacy score on the device.
//
// y_pred of mismatch.py:15-35 (calculate_y_pred) for N (original, variant) pairs of
// packed k-mers:  intercept + sum over differing positions (ascending) of
// [w_pos + w_{orig base, variant base}] + gc_weight * GC(original), in float64 with
// the reference's order of additions and no fused multiply-add, so the result is
// bit-identical to the Python float.  One lane per pair; the weight table
// (1 + 32 + 16 + 1 doubles) is read through the scalar / L1 path.
#include "bc_device.h"

using namespace bc;

namespace {

// weights layout: [0] intercept, [1..32] position 0..31, [33 + 4*o + v] substitution o->v
// (codes A C G T; the diagonal is unused), [49] GC_content
__global__ __launch_bounds__(BLOCK) void k_score(const uint64_t* __restrict__ orig, const uint64_t* __restrict__ var,
                                                 uint64_t n, int k, const double* __restrict__ w,
                                                 double* __restrict__ out, uint8_t* __restrict__ none) {
#pragma clang fp contract(off)  // hipcc contracts a*b+c into an FMA by default; Python rounds the product first
    const uint32_t km = (k >= 32) ? 0xFFFFFFFFu : ((1u << k) - 1u);
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint64_t a = orig[i], b = var[i];
        const uint32_t alo = (uint32_t)a & km, ahi = (uint32_t)(a >> 32) & km;
        const uint32_t blo = (uint32_t)b & km, bhi = (uint32_t)(b >> 32) & km;
        uint32_t mm = (alo ^ blo) | (ahi ^ bhi);
        const bool same = (mm == 0u);
        double y = w[0];
        while (mm) {  // ascending positions, two additions each (mismatch.py:29-32)
            const int p = __ffs(mm) - 1;
            mm &= mm - 1;
            const uint32_t o = ((alo >> p) & 1u) | (((ahi >> p) & 1u) << 1);
            const uint32_t v = ((blo >> p) & 1u) | (((bhi >> p) & 1u) << 1);
            y = __dadd_rn(y, w[1 + p]);
            y = __dadd_rn(y, w[33 + 4 * o + v]);
        }
        // GC content of the original: C = 01, G = 10  ->  lo ^ hi
        const double gc = __ddiv_rn((double)__popc(alo ^ ahi), (double)k);
        double prod = __dmul_rn(w[49], gc);
        asm volatile("" : "+v"(prod));  // keep the rounded product: no contraction into an FMA with the add below
        y = __dadd_rn(y, prod);
        out[i] = y;
        if (none) none[i] = same ? 1 : 0;  // the reference returns None for identical sequences
    }
}

}  // namespace

extern "C" {

int bc_score_mismatch_dev(const uint64_t* d_orig, const uint64_t* d_var, uint64_t n, int k, const double* d_weights,
                          double* d_out, uint8_t* d_none, void* stream) {
    if ((n && (!d_orig || !d_var || !d_out)) || !d_weights || k < 1 || k > 32) {
        set_error("bc_score_mismatch_dev: bad argument (k=%d)", k);
        return BC_EINVAL;
    }
    if (n == 0) return BC_OK;
    uint64_t b = (n + BLOCK - 1) / BLOCK;
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    hipLaunchKernelGGL(k_score, dim3((int)b), dim3(BLOCK), 0, (hipStream_t)stream, d_orig, d_var, n, k, d_weights,
                       d_out, d_none);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

}  // extern "C"
