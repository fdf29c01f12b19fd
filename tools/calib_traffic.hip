// calib_traffic.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS path's access
// pattern on gfx950: 8 bytes per lane, 512-byte contiguous row segments per wave-instruction
// (the k_render epilogue reads nelec and writes lambda exactly like this).
// MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for 16-B/lane streams (reads 1/2);
// "calibrate on a known byte count in your own access pattern before trusting an absolute".
//
//   hipcc --offload-arch=gfx950 -O3 -o calib_traffic tools/calib_traffic.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./calib_traffic
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out -- ./calib_traffic
// Known bytes per launch are printed; divide by the counter (KB) to get the correction.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
    } while (0)

// one wave per 64 x 32 tile of a W-wide image, like k_render's epilogue
__global__ void __launch_bounds__(64) k_read8(const double *__restrict__ in, double *__restrict__ out, int W, int ntx) {
    int tile = blockIdx.x, lane = threadIdx.x;
    int ty = tile / ntx, tx = tile - ty * ntx;
    const double *p = in + (size_t)(ty * 32) * W + tx * 64 + lane;
    double s = 0.0;
#pragma unroll 4
    for (int r = 0; r < 32; r++) s += p[(size_t)r * W];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) out[tile] = s;
}

__global__ void __launch_bounds__(64) k_write8(double *__restrict__ outp, int W, int ntx) {
    int tile = blockIdx.x, lane = threadIdx.x;
    int ty = tile / ntx, tx = tile - ty * ntx;
    double *p = outp + (size_t)(ty * 32) * W + tx * 64 + lane;
#pragma unroll 4
    for (int r = 0; r < 32; r++) p[(size_t)r * W] = (double)(r + lane);
}

int main() {
    // 2048 x 65536 doubles = 1 GiB: beyond the 256 MiB Infinity Cache
    const int W = 2048;
    const size_t H = 65536;
    const int ntx = W / 64;
    const int tiles = (int)(ntx * (H / 32));
    double *buf, *out;
    CK(hipMalloc(&buf, sizeof(double) * W * H));
    CK(hipMalloc(&out, sizeof(double) * tiles));
    CK(hipMemset(buf, 0, sizeof(double) * W * H));
    for (int it = 0; it < 3; it++) {
        hipLaunchKernelGGL(k_read8, dim3(tiles), dim3(64), 0, 0, buf, out, W, ntx);
        hipLaunchKernelGGL(k_write8, dim3(tiles), dim3(64), 0, 0, buf, W, ntx);
    }
    CK(hipDeviceSynchronize());
    printf("k_read8 : %zu bytes read per launch (%.1f KB)\n", sizeof(double) * W * H, sizeof(double) * W * H / 1024.0);
    printf("k_write8: %zu bytes written per launch (%.1f KB)\n", sizeof(double) * W * H, sizeof(double) * W * H / 1024.0);
    // the same pattern on a 168 MB buffer (5 x 2048 x 2048 doubles), read twice in a row: what an
    // image that fits the Infinity Cache looks like to the counter
    const size_t H2 = 5 * 2048;
    const int tiles2 = (int)(ntx * (H2 / 32));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_read8, dim3(tiles2), dim3(64), 0, 0, buf, out, W, ntx);
    CK(hipDeviceSynchronize());
    printf("k_read8 (small): %zu bytes read per launch (%.1f KB), launches 7..9\n", sizeof(double) * W * H2,
           sizeof(double) * W * H2 / 1024.0);
    return 0;
}
