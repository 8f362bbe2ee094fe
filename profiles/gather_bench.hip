// gather_bench.hip — what the MI355X gives a table-lookup kernel: random-gather
// rates (items/s) for the access shapes the matchers use, over table sizes
// from L2-resident to HBM-resident.  Build + run: profiles/gather_bench.sh.
//
// shapes (one "item" = one table visit of one read):
//   A  1 lane/item, one 16-B load            (hash probe of k_match_hash)
//   B  4 lanes/item, 16 B each = one 64-B slab piece (phase 2 of k_match_fused)
//   C  1 lane/item, 4 x 16 B from the same 64-B slab (single-lane slab scan)
//   D  1 lane/item, one 8-B load             (CSR gathers of k_match_seed)
//   E  1 lane/item, one 4-B load
// ILP = independent items in flight per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int SHAPE, int ILP>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ table, uint64_t slab_mask,
                                                uint64_t n_items, uint32_t* __restrict__ sink) {
    // table is addressed in 64-B slabs; slab_mask = n_slabs - 1
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    if (SHAPE == 1) {  // B: quad per item
        const uint64_t q = tid >> 2, nq = nthreads >> 2;
        const int sub = tid & 3;
        for (uint64_t i = q; i < n_items; i += nq * ILP) {
            uint4 v[ILP];
#pragma unroll
            for (int j = 0; j < ILP; ++j) {
                uint64_t it = i + (uint64_t)j * nq;
                uint64_t s = mix(it) & slab_mask;
                v[j] = (it < n_items) ? *(const uint4*)(table + s * 16 + sub * 4) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < ILP; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
    } else {
        for (uint64_t i = tid; i < n_items; i += nthreads * ILP) {
            if (SHAPE == 0) {
                uint4 v[ILP];
#pragma unroll
                for (int j = 0; j < ILP; ++j) {
                    uint64_t it = i + (uint64_t)j * nthreads;
                    uint64_t h = mix(it);
                    uint64_t s = h & slab_mask;
                    v[j] = (it < n_items) ? *(const uint4*)(table + s * 16 + ((h >> 60) & 3) * 4) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < ILP; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
            } else if (SHAPE == 2) {
                uint4 v[ILP][4];
#pragma unroll
                for (int j = 0; j < ILP; ++j) {
                    uint64_t it = i + (uint64_t)j * nthreads;
                    uint64_t s = mix(it) & slab_mask;
                    const uint4* p = (const uint4*)(table + s * 16);
                    bool ok = it < n_items;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[j][c] = ok ? p[c] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < ILP; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc ^= v[j][c].x ^ v[j][c].y ^ v[j][c].z ^ v[j][c].w;
            } else if (SHAPE == 3) {
                uint2 v[ILP];
#pragma unroll
                for (int j = 0; j < ILP; ++j) {
                    uint64_t it = i + (uint64_t)j * nthreads;
                    uint64_t h = mix(it);
                    uint64_t s = h & slab_mask;
                    v[j] = (it < n_items) ? *(const uint2*)(table + s * 16 + ((h >> 60) & 7) * 2) : make_uint2(0, 0);
                }
#pragma unroll
                for (int j = 0; j < ILP; ++j) acc ^= v[j].x ^ v[j].y;
            } else {
                uint32_t v[ILP];
#pragma unroll
                for (int j = 0; j < ILP; ++j) {
                    uint64_t it = i + (uint64_t)j * nthreads;
                    uint64_t h = mix(it);
                    uint64_t s = h & slab_mask;
                    v[j] = (it < n_items) ? table[s * 16 + ((h >> 60) & 15)] : 0u;
                }
#pragma unroll
                for (int j = 0; j < ILP; ++j) acc ^= v[j];
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;  // keep the loads alive
}

// S: the exact-match skeleton — stream an 8-B key in (4 per lane, 32-B loads), gather 16 B at an address derived
// from it, stream 4 B + 1 B out (int4 / uchar4 stores).  NT = non-temporal streams.
template <bool NT>
__global__ __launch_bounds__(256) void k_stream_gather(const uint32_t* __restrict__ table, uint64_t slab_mask,
                                                       const uint64_t* __restrict__ keys, uint64_t n_items,
                                                       int32_t* __restrict__ out_idx, uint8_t* __restrict__ out_d) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i * 4 < n_items; i += nthreads) {
        const ulonglong4* kp = (const ulonglong4*)keys + i;
        ulonglong4 k4;
        if (NT) {
            k4.x = __builtin_nontemporal_load(&kp->x); k4.y = __builtin_nontemporal_load(&kp->y);
            k4.z = __builtin_nontemporal_load(&kp->z); k4.w = __builtin_nontemporal_load(&kp->w);
        } else k4 = *kp;
        uint64_t k[4] = {k4.x, k4.y, k4.z, k4.w};
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t h = mix(k[j]);
            v[j] = *(const uint4*)(table + (h & slab_mask) * 16 + ((h >> 60) & 3) * 4);
        }
        int4 r; uchar4 d;
        r.x = v[0].x ^ v[0].z; r.y = v[1].x ^ v[1].z; r.z = v[2].x ^ v[2].z; r.w = v[3].x ^ v[3].z;
        d.x = v[0].y; d.y = v[1].y; d.z = v[2].y; d.w = v[3].y;
        if (NT) {
            int4* op = (int4*)out_idx + i;
            __builtin_nontemporal_store(r.x, &op->x); __builtin_nontemporal_store(r.y, &op->y);
            __builtin_nontemporal_store(r.z, &op->z); __builtin_nontemporal_store(r.w, &op->w);
            __builtin_nontemporal_store(*(uint32_t*)&d, (uint32_t*)out_d + i);
        } else {
            ((int4*)out_idx)[i] = r;
            ((uchar4*)out_d)[i] = d;
        }
    }
}

__global__ void k_fill_keys(uint64_t* keys, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = mix(i * 0x1234567ull + 99);
}

template <bool NT>
static void run_stream(const uint32_t* table, uint64_t bytes, const uint64_t* keys, uint64_t n_items, int32_t* oi,
                       uint8_t* od, int blocks) {
    uint64_t slab_mask = bytes / 64 - 1;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_stream_gather<NT><<<blocks, 256>>>(table, slab_mask, keys, n_items, oi, od);
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) k_stream_gather<NT><<<blocks, 256>>>(table, slab_mask, keys, n_items, oi, od);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    double gi = n_items / (ms * 1e6);
    printf("{\"shape\": \"S:stream8B+16Bgather+5Bout%s\", \"ilp\": 4, \"table_mib\": %.2f, \"blocks\": %d, \"ms\": %.3f, \"gitems_per_s\": %.1f, \"items_per_clk_cu\": %.3f}\n",
           NT ? "/nt" : "", bytes / 1048576.0, blocks, ms, gi, gi / (256 * 2.4));
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

template <int SHAPE, int ILP>
static void run(const char* name, const uint32_t* table, uint64_t bytes, uint64_t n_items, uint32_t* sink, int blocks) {
    uint64_t slab_mask = bytes / 64 - 1;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_gather<SHAPE, ILP><<<blocks, 256>>>(table, slab_mask, n_items / 8, sink);  // warm
    CK(hipEventRecord(a));
    k_gather<SHAPE, ILP><<<blocks, 256>>>(table, slab_mask, n_items, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    double gi = n_items / (ms * 1e6);
    printf("{\"shape\": \"%s\", \"ilp\": %d, \"table_mib\": %.0f, \"blocks\": %d, \"ms\": %.3f, \"gitems_per_s\": %.1f, \"items_per_clk_cu\": %.3f}\n",
           name, ILP, bytes / 1048576.0, blocks, ms, gi, gi / (256 * 2.4));
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main(int argc, char** argv) {
    const uint64_t n_items = argc > 1 ? strtoull(argv[1], 0, 10) : 400000000ull;
    const uint64_t max_bytes = 1ull << 31;
    uint32_t *table, *sink;
    CK(hipMalloc(&table, max_bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(table, 1, max_bytes));
    if (argc > 2) {  // stream + gather mode
        const uint64_t n = 100000000ull;
        uint64_t* keys; int32_t* oi; uint8_t* od;
        CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&oi, n * 4)); CK(hipMalloc(&od, n));
        k_fill_keys<<<(unsigned)((n + 255) / 256), 256>>>(keys, n);
        CK(hipDeviceSynchronize());
        for (uint64_t bytes : {1ull << 19, 1ull << 20, 2ull << 20, 4ull << 20, 8ull << 20, 32ull << 20})
            for (int blocks : {2048, 12208, 97657}) {
                run_stream<false>(table, bytes, keys, n, oi, od, blocks);
                run_stream<true>(table, bytes, keys, n, oi, od, blocks);
            }
        return 0;
    }
    const uint64_t sizes[] = {1ull << 20, 4ull << 20, 8ull << 20, 32ull << 20, 256ull << 20, 2048ull << 20};
    for (uint64_t bytes : sizes) {
        for (int blocks : {2048, 16384}) {
            run<0, 1>("A:16B/lane", table, bytes, n_items, sink, blocks);
            run<0, 4>("A:16B/lane", table, bytes, n_items, sink, blocks);
            run<0, 8>("A:16B/lane", table, bytes, n_items, sink, blocks);
            run<1, 1>("B:quad64B", table, bytes, n_items, sink, blocks);
            run<1, 4>("B:quad64B", table, bytes, n_items, sink, blocks);
            run<2, 1>("C:lane4x16B", table, bytes, n_items, sink, blocks);
            run<2, 2>("C:lane4x16B", table, bytes, n_items, sink, blocks);
            run<3, 4>("D:8B/lane", table, bytes, n_items, sink, blocks);
            run<4, 4>("E:4B/lane", table, bytes, n_items, sink, blocks);
        }
    }
    CK(hipFree(table)); CK(hipFree(sink));
    return 0;
}
