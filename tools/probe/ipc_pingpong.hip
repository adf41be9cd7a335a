// Feasibility probe (diagnostic, not part of the product): two PROCESSES on one GPU share a fine-grained buffer through
// hipIpc handles and run concurrent kernels that hand a word back and forth with system-scope stores / loads and
// bounded polls.  Prints the round-trip time.  Build: hipcc --offload-arch=gfx950 ipc_pingpong.hip -o ipc_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s -> %s\n", (int)getpid(), #x, hipGetErrorString(e_)); exit(2); } } while (0)

// buf[0]: word written by rank 0, buf[32]: word written by rank 1, buf[64 + r]: result (cycles), buf[96]: timeout flag
__global__ void pingpong(unsigned* mine, unsigned* theirs, unsigned* res, unsigned* tmo, int rank, int iters) {
    if (threadIdx.x != 0) return;
    long long t0 = 0;
    for (int i = 1; i <= iters; ++i) {
        if (i == 2) t0 = wall_clock64();
        if (rank == 0) __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned spins = 0;
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned)i) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24) || (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u && spins > 1000u)) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                res[1] = (unsigned)i;
                return;
            }
        }
        if (rank == 1) __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    res[0] = (unsigned)((wall_clock64() - t0) / (iters - 1));   // 100 MHz ticks per round trip
    res[1] = 0;
}

int main(int argc, char** argv) {
    const int finegrained = argc > 1 ? atoi(argv[1]) : 1;
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    pid_t pid = fork();                      // before any HIP call
    const int rank = pid == 0 ? 1 : 0;
    CK(hipSetDevice(0));
    unsigned* buf = nullptr;
    hipIpcMemHandle_t hnd;
    if (rank == 0) {
        if (finegrained) CK(hipExtMallocWithFlags((void**)&buf, 4096, hipDeviceMallocFinegrained));
        else CK(hipMalloc((void**)&buf, 4096));
        CK(hipMemset(buf, 0, 4096));
        CK(hipDeviceSynchronize());
        CK(hipIpcGetMemHandle(&hnd, buf));
        if (write(p2c[1], &hnd, sizeof hnd) != (ssize_t)sizeof hnd) return 1;
    } else {
        if (read(p2c[0], &hnd, sizeof hnd) != (ssize_t)sizeof hnd) return 1;
        CK(hipIpcOpenMemHandle((void**)&buf, hnd, hipIpcMemLazyEnablePeerAccess));
    }
    char go = 1;
    if (rank == 1) { if (write(c2p[1], &go, 1) != 1) return 1; } else { if (read(c2p[0], &go, 1) != 1) return 1; }
    unsigned* mine = buf + (rank == 0 ? 0 : 32);
    unsigned* theirs = buf + (rank == 0 ? 32 : 0);
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, mine, theirs, buf + 64 + 2 * rank, buf + 96, rank, 2000);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    unsigned res[2];
    CK(hipMemcpy(res, buf + 64 + 2 * rank, sizeof res, hipMemcpyDeviceToHost));
    printf("rank %d (finegrained=%d): %s, round trip %.2f us (stuck at iteration %u)\n", rank, finegrained, res[1] ? "TIMEOUT" : "ok", res[0] / 100.0, res[1]);
    fflush(stdout);
    if (rank == 1) { CK(hipIpcCloseMemHandle(buf)); return res[1] ? 3 : 0; }
    int st = 0;
    waitpid(pid, &st, 0);
    CK(hipFree(buf));
    return (res[1] || st) ? 3 : 0;
}
