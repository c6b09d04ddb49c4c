// one-off: HBM write rate of a [M][N] bf16 matrix (x2 planes) as a function of the row-segment width one store instruction covers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)
// SEG = bytes per row segment (64, 128, 256, 512); each lane stores 16 B; 512 threads = 8 waves; persistent: 256 WGs walk tiles of 128 rows x 768 B
template <int SEG>
__global__ __launch_bounds__(512) void wr(uint4* out, long plane_u4, int M, int row_u4, int ntiles_n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16;           // lanes per row segment
    constexpr int RPI = 64 / LPR;           // rows per instruction
    const int nbm = (M + 127) / 128;
    const uint4 v = {1u, 2u, 3u, (unsigned)lane};
    for (int t = blockIdx.x; t < nbm * ntiles_n; t += gridDim.x) {
        const int bm = t / ntiles_n, bn = t % ntiles_n;
        // the wave's share: 64 rows (wr) x 192 B (wc) of the 128 x 768 B tile
        const int wr_ = wave >> 2, wc = wave & 3;
        // 64 rows x 192 B = 12 KiB = 12 instructions of 1 KiB per plane
        for (int k = 0; k < 12; ++k) {
            int row, colb;
            if (SEG == 32) {            // 6 column blocks of 32 B, 2 row groups of 32
                const int j = k / 2, ps = k % 2;
                row = ps * 32 + lane / LPR; colb = wc * 192 + j * 32 + (lane % LPR) * 16;
            } else if (SEG <= 64) {            // 3 column blocks of 64 B, 4 row groups of 16
                const int j = k / 4, ps = k % 4;
                row = ps * 16 + lane / LPR; colb = wc * 192 + j * 64 + (lane % LPR) * 16;
            } else {                    // (timing only) the tile's 768 B rows are shared differently: wave = 16 rows x 768 B
                const int rows_per_wave = 16, seg_per_row = 768 / SEG;
                const int idx = k * RPI + lane / LPR;       // segment index inside the wave's 16 x 768 B
                row = wave * rows_per_wave + idx / seg_per_row - wr_ * 0;
                colb = (idx % seg_per_row) * SEG + (lane % LPR) * 16;
                row = row % 128;
            }
            int gm = bm * 128 + (SEG <= 64 ? wr_ * 64 : 0) + row;
            gm = gm < M ? gm : M - 1;
            uint4* dst = out + (long)gm * row_u4 + (bn * 768 + colb) / 16;
            dst[0] = v;
            dst[plane_u4] = v;
        }
    }
}
int main() {
    const int M = 115232, N = 1536;
    const long plane = (long)M * N * 2;
    uint4* out;
    CK(hipMalloc(&out, 2 * plane));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char* name) {
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, plane / 16, M, N * 2 / 16, N * 2 / 768);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s: %.1f us  %.2f TB/s\n", name, ms * 100, 2.0 * plane / (ms / 10 * 1e-3) / 1e12);
        }
    };
    run(wr<32>, "32 B segments (32 rows / instruction)");
    run(wr<64>, "64 B segments (16 rows / instruction)");
    run(wr<128>, "128 B segments");
    run(wr<256>, "256 B segments");
    run(wr<768>, "768 B rows");
    CK(hipDeviceSynchronize());
    return 0;
}
