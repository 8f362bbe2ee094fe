// Does a 64-byte sector that a wave has just read through the scalar cache hit in the L2 when the same wave reads it
// with vector loads?  One wave per 2 KB tile of a large buffer: (mode & 1) scalar-touch the tile first (32 x 4 bytes,
// one per 64-byte sector, awaited), then two 16-byte vector loads per lane — (mode & 2) non-temporal, (mode & 4) lane l
// at l*16 and 1024 + l*16 instead of l*32 and l*32 + 16.  Run under rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void k(const unsigned* __restrict__ buf, unsigned* __restrict__ out) {
    const unsigned* tile = buf + (size_t)blockIdx.x * 512;
    unsigned acc = 0;
    if (MODE & 1) {
        typedef const unsigned __attribute__((address_space(4)))* cu;
        cu p = (cu)(uintptr_t)tile;
#pragma unroll
        for (int i = 0; i < 32; ++i) acc ^= p[i * 16];
        asm volatile("" ::"s"(acc));
    }
    const unsigned lane = threadIdx.x;
    const u32x4* q = (const u32x4*)tile;
    u32x4 a, b;
    const unsigned i0 = (MODE & 4) ? lane : lane * 2, i1 = (MODE & 4) ? 64 + lane : lane * 2 + 1;
    if (MODE & 2) { a = __builtin_nontemporal_load(q + i0); b = __builtin_nontemporal_load(q + i1); }
    else { a = q[i0]; b = q[i1]; }
    unsigned v = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ acc;
    if (v == 0x12345678u) out[lane] = v;
}
int main(int argc, char** argv) {
    const size_t tiles = 390625;
    unsigned *d, *o;
    hipMalloc(&d, tiles * 2048); hipMalloc(&o, 4096);
    hipMemset(d, 1, tiles * 2048);
    hipDeviceSynchronize();
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(tiles), dim3(64), 0, 0, d, o); hipDeviceSynchronize();
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    printf("done\n");
    return 0;
}
