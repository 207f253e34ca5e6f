// How long does a wave wait for its stores on gfx950?  (hipcc --offload-arch=gfx950 -O3 wt_drain.hip)
//   per iteration: one 16-byte store per lane (1 KB per wave) followed by s_waitcnt vmcnt(0); time = kernel duration / iterations
//   (all waves run concurrently; the s_memtime ticks are printed as well).  Variants: plain store, sc1 (write-through to the memory side: the hand-off
//   payload of the sequence kernels), sc0 sc1; and an L2-hit load round trip for scale.  256 work-groups x 4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *buf, long long *cyc, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float *p = buf + (size_t)tid * 4;
    f32x4 v = {1.f, 2.f, 3.f, (float)tid};
    const size_t stride = (size_t)gridDim.x * 256 * 4;  // a fresh line every iteration
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        float *q = p + (size_t)(i & 63) * stride;
        if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" ::"v"(q), "v"(v) : "memory");
        if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" ::"v"(q), "v"(v) : "memory");
        if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" ::"v"(q), "v"(v) : "memory");
        if (MODE == 3) {
            f32x4 r;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(q) : "memory");
            v[0] += r[1];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = (t1 - t0) / iters;
    if (v[0] == 123456.f) buf[0] = v[0];
}

template <int MODE>
void run(const char *name, float *buf, long long *cyc, int grid) {
    const int iters = 2000;
    k<MODE><<<grid, 256>>>(buf, cyc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(buf, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[1024];
    (void)hipMemcpy(h, cyc, sizeof(long long) * grid * 4, hipMemcpyDeviceToHost);
    double s = 0;
    long long mx = 0;
    for (int i = 0; i < grid * 4; ++i) { s += (double)h[i]; mx = h[i] > mx ? h[i] : mx; }
    printf("%-34s grid %3d: %6.2f us per op+wait (kernel time / iterations); counter: mean %7.0f, slowest wave %lld ticks\n", name,
           grid, ms * 1e3 / iters, s / (grid * 4), mx);
}

int main() {
    float *buf;
    long long *cyc;
    (void)hipMalloc(&buf, (size_t)256 * 256 * 4 * 4 * 64);
    (void)hipMalloc(&cyc, sizeof(long long) * 1024);
    for (int grid : {1, 256}) {
        run<0>("plain store", buf, cyc, grid);
        run<1>("sc1 store (write-through)", buf, cyc, grid);
        run<2>("sc0 sc1 store", buf, cyc, grid);
        run<3>("sc1 load (L2 / memory round trip)", buf, cyc, grid);
    }
    return 0;
}
