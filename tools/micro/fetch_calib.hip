// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the fused kernel's access patterns (MI355X_MICROARCH.md, HBM
// section: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   k_rows16   : the pair-row pattern of epnn_wave.hip.h (load_e1 / the front-end's pt store): lane (q, n16) moves the
//                16 bytes at row[n16-th row of the tile] + 16 q of 64-byte rows, rows visited in a scattered order
//   k_frag     : the weight-fragment pattern (W16_LD): one dword per lane, 256 contiguous bytes per wavefront and step
//   k_stream16 : 16 bytes per lane, fully coalesced (the guide's reference case: FETCH_SIZE reads 1/2)
// Every kernel reads (and k_rows16 also writes) every byte of its buffer exactly once; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- ./fetch_calib      (and again with WRITE_SIZE)
// and compare the counters (KiB) with the byte counts printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_rows16(const float *src, float *dst, const int *perm, int tiles_per_wave, float *sink) {
    const int lane = threadIdx.x, q = lane >> 4, n16 = lane & 15;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int tile = perm[blockIdx.x * tiles_per_wave + t];          // 16 rows of 64 bytes
        const size_t off = ((size_t)tile * 16 + n16) * 16 + 4 * q;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + off);
        acc += v;
        *reinterpret_cast<f32x4 *>(dst + off) = v * 2.f;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
__global__ __launch_bounds__(64) void k_frag(const float *src, int steps, float *sink) {
    const int lane = threadIdx.x;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) acc += src[((size_t)blockIdx.x * steps + s) * 64 + lane];
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream16(const float *src, size_t n4, float *sink) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        acc += reinterpret_cast<const f32x4 *>(src)[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main() {
    const size_t bytes = 512ull << 20;                       // beyond the 256 MiB Infinity Cache
    const size_t nfl = bytes / 4;
    float *src, *dst, *sink;
    int *perm;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(src, 0, bytes));
    CHECK(hipMemset(dst, 0, bytes));
    const int ntiles = (int)(bytes / 1024), tpw = 64, nwaves = ntiles / tpw;
    std::vector<int> p(ntiles);
    for (int i = 0; i < ntiles; ++i) p[i] = (int)(((long long)i * 1000003LL) % ntiles);   // a permutation: 1000003 is prime, ntiles a power of two
    CHECK(hipMalloc(&perm, (size_t)ntiles * 4));
    CHECK(hipMemcpy(perm, p.data(), (size_t)ntiles * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_rows16, dim3(nwaves), dim3(64), 0, 0, src, dst, perm, tpw, sink);
        hipLaunchKernelGGL(k_frag, dim3((unsigned)(nfl / (64 * 256))), dim3(64), 0, 0, src, 256, sink);
        hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, src, nfl / 4, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("bytes read by every kernel: %zu (%.1f KiB); k_rows16 also writes as many\n", bytes, bytes / 1024.0);
    return 0;
}
