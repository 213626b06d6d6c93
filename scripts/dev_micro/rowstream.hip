// Read-bandwidth probe (development aid): 8 row streams of 128 B per wave-load against 1 KiB contiguous per wave-load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int COLS = 576, ROWB = COLS * 128;
template <int MODE, int UNR>
__global__ void __launch_bounds__(64) k_read(const char *__restrict__ vol, uint32_t *out, int sweeps)
{
    const int tid = threadIdx.x, g = tid >> 3, l = tid & 7;
    const char *base = vol + (size_t)blockIdx.x * 8 * ROWB;
    u32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < sweeps; s++) {
        if (MODE == 0) {
            const char *p = base + (size_t)g * ROWB + l * 16;
            for (int c = 0; c < COLS; c += UNR) {
                u32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; u++) { const int cc = s & 1 ? COLS - 1 - (c + u) : c + u; v[u] = *reinterpret_cast<const u32x4 *>(p + cc * 128); }
#pragma unroll
                for (int u = 0; u < UNR; u++) acc ^= v[u];
            }
        } else {
            // same bytes, 1 KiB contiguous per wave-load: 8 columns of one row, row after row
            const char *p = base + tid * 16;
            for (int cb = 0; cb < COLS / 8; cb += UNR / 8) {
                u32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; u++) { const int r = u & 7, blk = cb + (u >> 3); const int bb = s & 1 ? COLS / 8 - 1 - blk : blk; v[u] = *reinterpret_cast<const u32x4 *>(p + (size_t)r * ROWB + (size_t)bb * 1024); }
#pragma unroll
                for (int u = 0; u < UNR; u++) acc ^= v[u];
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;
}
int main()
{
    const int nblk = 11520;
    const size_t bytes = (size_t)nblk * 8 * ROWB;
    char *vol; uint32_t *out;
    hipMalloc(&vol, bytes); hipMalloc(&out, nblk * 4);
    hipMemset(vol, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0); kern<<<nblk, 64>>>(vol, out, 2); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-28s %.3f ms  %.2f TB/s\n", name, best, 2.0 * bytes / best / 1e9);
    };
    printf("volume %.2f GB\n", bytes / 1e9);
    run("rows x128B, 8 in flight", k_read<0, 8>);
    run("rows x128B, 16 in flight", k_read<0, 16>);
    run("rows x128B, 32 in flight", k_read<0, 32>);
    run("1KiB contiguous, 8", k_read<1, 8>);
    run("1KiB contiguous, 16", k_read<1, 16>);
    run("1KiB contiguous, 32", k_read<1, 32>);
    return 0;
}
