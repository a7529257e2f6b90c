// Feasibility probe for the gru rendezvous (round 6): can a stream wait for a 32-bit value that ANOTHER stream writes later (host order: the wait is
// enqueued first)?  hipStreamWaitValue32 on signal memory (hipExtMallocWithFlags(.., hipMallocSignalMemory)) + hipStreamWriteValue32.
//   build: hipcc --offload-arch=gfx950 -O2 tools/micro/stream_wait_lab.hip -o tools/micro/bin/stream_wait_lab;  run under `timeout 60`
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void stamp(unsigned long long* out, int slot, int spin) {
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    out[slot] = t;
}
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    uint32_t* flag[3];
    for (int i = 0; i < 3; ++i) { CK(hipExtMallocWithFlags((void**)&flag[i], 8, hipMallocSignalMemory)); CK(hipMemset(flag[i], 0, 8)); }
    unsigned long long* out; CK(hipMalloc(&out, 64)); CK(hipMemset(out, 0, 64));
    hipStream_t s[3]; for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    for (unsigned epoch = 1; epoch <= 3; ++epoch) {
        // each stream: "pre" kernel (long on stream 2), write own flag, wait for the two others, "recurrence" stamp
        for (int i = 0; i < 3; ++i) {
            hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s[i], out, 4, i == 2 ? 20000 : 10);      // pre-work; stream 2's takes ~ms
            CK(hipStreamWriteValue32(s[i], flag[i], epoch, 0));
            for (int j = 0; j < 3; ++j) if (j != i) CK(hipStreamWaitValue32(s[i], flag[j], epoch, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s[i], out, i, 0);
            usleep(300);                                                                       // the host enqueues the passes one after the other
        }
        for (int i = 0; i < 3; ++i) CK(hipStreamSynchronize(s[i]));
        unsigned long long h[8]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
        printf("epoch %u: recurrence start stamps relative to stream 0 (100 MHz ticks): %lld %lld %lld   (pre-work of stream 2 started at %lld)\n", epoch,
               0ll, (long long)(h[1] - h[0]), (long long)(h[2] - h[0]), (long long)(h[4] - h[0]));
    }
    printf("ok\n");
    return 0;
}
