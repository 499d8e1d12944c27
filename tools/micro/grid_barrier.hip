// Microbenchmark: what a device-wide barrier inside ONE launch costs against a kernel boundary (the training step of one
// molecule is 21 dependent launches of 41-246 workgroups).  A persistent grid of G workgroups x 512
// threads (cooperative launch: all resident) runs K phases; every phase each workgroup writes a row another workgroup reads in
// the next one (so the barrier must really publish memory: agent-scope release before the arrival, acquire after the wait),
// then crosses the barrier: __syncthreads, thread 0 does fence + atomicAdd + a bounded spin on the counter + fence.
// Printed: microseconds per phase for the barrier form and for the same K phases as K dependent launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool grid_sync(unsigned *ctr, unsigned target) {
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(ctr, 1u);
        int good = 1;
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) { good = 0; break; }          // a bounded wait: never a hung grid
        }
        __threadfence();
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__device__ __forceinline__ void phase_body(float *buf, int G, int wg, int ph) {
    // read the row the neighbour wrote in the previous phase, write this workgroup's row
    const float *src = buf + (size_t)((ph & 1) ^ 1) * G * 512 + (size_t)((wg + 1) % G) * 512;
    float *dst = buf + (size_t)(ph & 1) * G * 512 + (size_t)wg * 512;
    dst[threadIdx.x] = src[threadIdx.x] + 1.f;
}

__global__ __launch_bounds__(512) void k_persist(float *buf, unsigned *ctr, int K, int *status) {
    const int G = gridDim.x, wg = blockIdx.x;
    for (int ph = 0; ph < K; ++ph) {
        phase_body(buf, G, wg, ph);
        if (!grid_sync(ctr, (unsigned)(ph + 1) * G)) { if (threadIdx.x == 0) *status = 1; return; }
    }
}
__global__ __launch_bounds__(512) void k_phase(float *buf, int ph) { phase_body(buf, gridDim.x, blockIdx.x, ph); }

int main(int argc, char **argv) {
    const int K = 40;
    for (int G : {41, 164, 246}) {
        float *buf; unsigned *ctr; int *status;
        CHK(hipMalloc(&buf, (size_t)2 * G * 512 * 4));
        CHK(hipMemset(buf, 0, (size_t)2 * G * 512 * 4));
        CHK(hipMalloc(&ctr, 4));
        CHK(hipMalloc(&status, 4));
        hipStream_t st; CHK(hipStreamCreate(&st));
        double best[2] = {1e9, 1e9};
        for (int rep = 0; rep < 6; ++rep) {
            CHK(hipMemsetAsync(ctr, 0, 4, st));
            CHK(hipMemsetAsync(status, 0, 4, st));
            CHK(hipStreamSynchronize(st));
            int k = K;
            void *args[] = {&buf, &ctr, &k, &status};
            auto t0 = std::chrono::steady_clock::now();
            CHK(hipLaunchCooperativeKernel((const void *)k_persist, dim3(G), dim3(512), args, 100 * 1024, st));
            CHK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) best[0] = std::min(best[0], us);
            int hs = 0; CHK(hipMemcpy(&hs, status, 4, hipMemcpyDeviceToHost));
            if (hs) { printf("G=%d: barrier wait ran out\n", G); return 1; }
            t0 = std::chrono::steady_clock::now();
            for (int ph = 0; ph < K; ++ph) hipLaunchKernelGGL(k_phase, dim3(G), dim3(512), 100 * 1024, st, buf, ph);
            CHK(hipStreamSynchronize(st));
            us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) best[1] = std::min(best[1], us);
        }
        std::vector<float> h((size_t)G * 512);
        CHK(hipMemcpy(h.data(), buf + (size_t)((K - 1) & 1) * G * 512, h.size() * 4, hipMemcpyDeviceToHost));
        printf("G = %3d workgroups x 512 threads (100 KB LDS each), %d phases: one persistent launch %.1f us = %.2f us per phase (barrier); "
               "%d dependent launches %.1f us = %.2f us per phase; check %.0f (expect %d)\n",
               G, K, best[0], best[0] / K, K, best[1], best[1] / K, h[0], 2 * K);
        (void)hipFree(buf); (void)hipFree(ctr); (void)hipFree(status); (void)hipStreamDestroy(st);
    }
    return 0;
}
