// Microbenchmark: what HBM sustains for conv1's traffic MIX — per sample 28 224 B read (u8 frame) and 51 200 B written
// (20 x 20 x 32 f32), persistent work-groups, 16-byte loads / stores, nothing else — next to pure read and pure write of
// the same volume.  The conv1 pair's "fraction of 8 TB/s" is priced against the spec sheet; this is the ceiling a kernel
// with that read : write ratio can actually reach.      hipcc --offload-arch=gfx950 -O3 stream_rw.hip -o stream_rw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// MODE 0: read + write (conv1's mix), 1: read only, 2: write only;  SEG: bytes of contiguous output one wave-store covers per
// pixel line (128 = whole 128-byte lines by one wave, 64 = two waves write the halves of every line at different times)
template <int MODE, int SEG>
__global__ __launch_bounds__(256, 2) void k_stream(const uint4 *__restrict__ in, uint4 *__restrict__ out, int n, unsigned *sink) {
    constexpr int RQ = 28224 / 16, WQ = 51200 / 16;  // 16-byte words per sample
    const int tid = threadIdx.x;
    unsigned acc = 0;
    for (int s = blockIdx.x; s < n; s += gridDim.x) {
        uint4 v = {1u, 2u, 3u, 4u};
        if (MODE != 2) {
            const uint4 *p = in + (size_t)s * RQ;
            for (int q = tid; q < RQ; q += 256) {
                const uint4 t = p[q];
                v.x ^= t.x; v.y ^= t.y; v.z ^= t.z; v.w ^= t.w;
            }
        }
        if (MODE != 1) {
            uint4 *p = out + (size_t)s * WQ;
            if (SEG == 128) {
                for (int q = tid; q < WQ; q += 256) p[q] = v;
            } else {  // wave w writes the 64-byte half (w & 1) of the lines: lanes 0..3 = 64 B of line 0, 4..7 of line 1, ...
                const int wave = tid >> 6, lane = tid & 63, half = wave & 1, grp = wave >> 1;
                for (int line = grp * 16 + (lane >> 2); line < WQ / 8; line += 32) p[line * 8 + half * 4 + (lane & 3)] = v;
            }
        } else {
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (MODE == 1 && acc == 0x12345678u) *sink = acc;
}

template <int MODE, int SEG>
static void run(const char *what, const uint4 *in, uint4 *out, int n, unsigned *sink, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        k_stream<MODE, SEG><<<dim3(grid), dim3(256)>>>(in, out, n, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)n * ((MODE != 2 ? 28224.0 : 0.0) + (MODE != 1 ? 51200.0 : 0.0));
    printf("%-52s grid %5d: %8.1f us  %6.2f TB/s\n", what, grid, best * 1e3, bytes / best / 1e9);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32768;
    uint4 *in, *out;
    unsigned *sink;
    hipMalloc(&in, (size_t)n * 28224);
    hipMalloc(&out, (size_t)n * 51200);
    hipMalloc(&sink, 4);
    hipMemset(in, 1, (size_t)n * 28224);
    printf("n = %d samples: %.2f GB read, %.2f GB written per pass\n", n, n * 28224.0 / 1e9, n * 51200.0 / 1e9);
    for (int grid : {512, 1024, 4096}) {
        run<0, 128>("read 28 KB + write 51 KB per sample, whole lines", in, out, n, sink, grid);
        run<0, 64>("read 28 KB + write 51 KB per sample, 64-byte halves", in, out, n, sink, grid);
        run<1, 128>("read only", in, out, n, sink, grid);
        run<2, 128>("write only, whole lines", in, out, n, sink, grid);
        run<2, 64>("write only, 64-byte halves", in, out, n, sink, grid);
    }
    return 0;
}
