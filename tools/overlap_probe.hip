// Does a large host-to-device copy on one thread overlap kernels launched from another thread?  (the block pass: one caller's 50 MB
// upload against another caller's device phase).  Variants of the copy: synchronous hipMemcpy from pageable memory, hipMemcpyAsync on
// its own stream from pageable / registered / pinned memory, and the pageable copy cut into chunks.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/overlap_probe.hip -o /tmp/overlap_probe -lpthread && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

__global__ void spin_kernel(uint32_t* out, uint32_t iters) {
    uint32_t v = threadIdx.x;
    for (uint32_t i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 0xDEADBEEF) out[0] = v;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t N = 50u << 20;
    uint8_t* pageable = (uint8_t*)malloc(N);
    memset(pageable, 1, N);
    uint8_t* registered = (uint8_t*)malloc(N);
    memset(registered, 2, N);
    uint8_t* pinned = nullptr;
    hipHostMalloc((void**)&pinned, N, hipHostMallocDefault);
    memset(pinned, 3, N);
    void *d = nullptr, *dk = nullptr;
    hipMalloc(&d, N);
    hipMalloc(&dk, 4096);
    hipStream_t sc, sk;
    hipStreamCreateWithFlags(&sc, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sk, hipStreamNonBlocking);
    // calibrate the kernel to ~1 ms
    uint32_t iters = 200000;
    for (int k = 0; k < 3; k++) {
        double t0 = now_ms();
        hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, sk, (uint32_t*)dk, iters);
        hipStreamSynchronize(sk);
        double dt = now_ms() - t0;
        if (k) iters = (uint32_t)(iters * 1.0 / dt);
    }
    double t0 = now_ms();
    hipHostRegister(registered, N, hipHostRegisterDefault);
    printf("hipHostRegister of 50 MB: %.2f ms\n", now_ms() - t0);
    auto copy = [&](int mode) {
        switch (mode) {
            case 0: hipMemcpy(d, pageable, N, hipMemcpyHostToDevice); break;
            case 1: hipMemcpyAsync(d, pageable, N, hipMemcpyHostToDevice, sc); hipStreamSynchronize(sc); break;
            case 2: hipMemcpyAsync(d, registered, N, hipMemcpyHostToDevice, sc); hipStreamSynchronize(sc); break;
            case 3: hipMemcpyAsync(d, pinned, N, hipMemcpyHostToDevice, sc); hipStreamSynchronize(sc); break;
            case 4:
                for (size_t o = 0; o < N; o += N / 8) hipMemcpyAsync((uint8_t*)d + o, pageable + o, N / 8, hipMemcpyHostToDevice, sc);
                hipStreamSynchronize(sc);
                break;
            case 5: {   // register in place, copy, unregister (what a caller's buffer would need)
                hipHostRegister(pageable, N, hipHostRegisterDefault);
                hipMemcpyAsync(d, pageable, N, hipMemcpyHostToDevice, sc);
                hipStreamSynchronize(sc);
                hipHostUnregister(pageable);
                break;
            }
        }
    };
    const char* names[] = {"hipMemcpy pageable", "async pageable", "async registered", "async pinned", "async pageable x8 chunks", "register+copy+unregister"};
    for (int mode = 0; mode < 6; mode++) {
        copy(mode);
        double c0 = now_ms();
        for (int k = 0; k < 10; k++) copy(mode);
        const double copy_alone = (now_ms() - c0) / 10;
        double k0 = now_ms();
        for (int k = 0; k < 10; k++) {
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, sk, (uint32_t*)dk, iters);
            hipStreamSynchronize(sk);
        }
        const double kern_alone = (now_ms() - k0) / 10;
        // together: thread A copies 20 times, thread B runs kernels until A is done
        std::atomic<bool> done(false);
        std::atomic<int> kernels(0);
        double tb0 = now_ms();
        std::thread b([&] {
            while (!done.load()) {
                hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, sk, (uint32_t*)dk, iters);
                hipStreamSynchronize(sk);
                kernels++;
            }
        });
        double a0 = now_ms();
        for (int k = 0; k < 20; k++) copy(mode);
        const double copy_together = (now_ms() - a0) / 20;
        done.store(true);
        b.join();
        const double wall = now_ms() - tb0;
        printf("%-28s copy alone %.2f ms (%.1f GB/s), kernel alone %.2f ms; together: copy %.2f ms, kernel %.2f ms each (%d in %.1f ms)\n", names[mode], copy_alone,
               N / copy_alone / 1e6, kern_alone, copy_together, wall / kernels.load(), kernels.load(), wall);
    }
    return 0;
}
