// Micro-benchmark: how many cycles does the vector-memory front end (texture addresser + L1) spend on one wave-wide
// load instruction, as a function of width, alignment and of how many distinct rows (cache lines) its lanes touch?
// (the KLT kernels issue ~30 such loads per keypoint-wave and level pass).  Working set: L2-resident image rows.
//   build: hipcc -O3 --offload-arch=gfx950 -o build/ta_bench scripts/micro/ta_bench.hip ; run: build/ta_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int W> struct vec;
template <> struct vec<16> { typedef uint4 T; static __device__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; } };
template <> struct vec<8> { typedef uint2 T; static __device__ unsigned fold(uint2 v) { return v.x ^ v.y; } };
template <> struct vec<4> { typedef unsigned T; static __device__ unsigned fold(unsigned v) { return v; } };

// lanes_per_row lanes share a row (consecutive W-byte pieces); misalign = byte offset added to every address (multiple of 4)
template <int W>
__global__ __launch_bounds__(64) void k(const unsigned char *__restrict__ img, int stride, int rows, int lanes_per_row, int misalign,
                                        int iters, unsigned *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int r = lane / lanes_per_row, p = lane % lanes_per_row;
    unsigned acc = 0;
    unsigned seed = blockIdx.x * 2654435761u + 12345u;
    for (int it = 0; it < iters; it += 8) {
        typename vec<W>::T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            seed = seed * 1664525u + 1013904223u;                 // wave-uniform window position
            const int row0 = (int)((seed >> 8) % (unsigned)(rows - 64));
            const int col0 = (int)((seed >> 20) % (unsigned)((stride - 64 - W * lanes_per_row) / 16)) * 16;
            const unsigned char *a = img + (size_t)(row0 + r) * stride + col0 + p * W + misalign;
            v[u] = *reinterpret_cast<const typename vec<W>::T *>(__builtin_assume_aligned(a, 4));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= vec<W>::fold(v[u]);
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main()
{
    const int stride = 832, rows = 498 * 8;   // 3.3 MB: eight padded level-0 planes, L2-resident
    unsigned char *img; unsigned *out;
    const int blocks = 256 * 16, iters = 512;
    hipMalloc(&img, (size_t)stride * rows + 4096); hipMemset(img, 1, (size_t)stride * rows + 4096);
    hipMalloc(&out, blocks * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct cfg { int w, lpr, mis; };
    const cfg cfgs[] = {{16, 1, 0}, {16, 1, 4}, {16, 1, 8}, {16, 3, 0}, {16, 3, 4}, {16, 2, 0}, {16, 4, 0}, {16, 4, 4}, {16, 8, 0}, {16, 64, 0},
                        {8, 1, 0}, {8, 1, 4}, {8, 3, 0}, {8, 3, 4}, {8, 8, 0}, {4, 1, 0}, {4, 3, 0}, {4, 16, 0}};
    printf("width lanes/row misalign   us    cycles/instr/CU(2.4GHz)\n");
    for (const cfg &c : cfgs) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (c.w == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(64), 0, 0, img, stride, rows, c.lpr, c.mis, iters, out);
            else if (c.w == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(64), 0, 0, img, stride, rows, c.lpr, c.mis, iters, out);
            else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, img, stride, rows, c.lpr, c.mis, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double instr_per_cu = (double)blocks * iters / 256.0;
        printf("%5d %9d %8d %8.1f %10.1f\n", c.w, c.lpr, c.mis, best * 1e3, best * 1e-3 * 2.4e9 / instr_per_cu);
    }
    return 0;
}
