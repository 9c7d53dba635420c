// Minimal stand-alone reproducer of the HIP runtime defect behind DESIGN.md section 1 ("Stream capture"): nothing of this
// library is involved.  Topology captured from an origin stream O with two side streams A and B:
//
//     O: k -> eFork -------------------------------------------------> wait(eJoinA), wait(eJoinB) -> k -> EndCapture
//     A: wait(eFork) -> k -> eA1 ............ wait(eB1) -> k -> eJoinA
//     B: wait(eFork) -> wait(eA1) -> k -> eB1 -> k -> eJoinB
//
// i.e. B waits on an event recorded by A and, later, A waits on an event recorded by B - a legal DAG (CUDA's capture rules:
// fork from the capturing stream, any cross-stream event edges between streams of the same capture, everything joined back).
// (modes 3-5 add what the library's call has on top: 3 an eager pass over the same streams and events first, 4 lowest-priority
// side streams, 5 further waits of the side streams on origin events in between)
// Mode 0 leaves the A <-> B edges out (A and B only talk to O): capture, instantiate and launch succeed.  Mode 1 adds the
// A -> B edge only, mode 2 both edges.  On the runtimes tried (profiles/r05_capture_cycle_repro.txt) modes 0-4 all pass -
// mutual waits between two side streams are harmless by themselves - and MODE 5 (each side stream also waits on an
// origin-stream event again between the mutual waits, as this library's side streams do on every octave's event) never
// returns from hipStreamEndCapture: the process dies of stack exhaustion (SIGSEGV) inside libamdhip64.so.
//
//   hipcc --offload-arch=gfx950 -O2 tools/capture_cycle_repro.hip -o tools/capture_cycle_repro && tools/capture_cycle_repro <mode>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        std::printf("%-58s %s\n", #x, e_ == hipSuccess ? "hipSuccess" : hipGetErrorString(e_)); \
        std::fflush(stdout);                                                                   \
        if (e_ != hipSuccess) return 1;                                                        \
    } while (0)

__global__ void k_inc(int* p) { atomicAdd(p, 1); }

int main(int argc, char** argv) {
    const int mode = argc > 1 ? std::atoi(argv[1]) : 2;
    int ver = 0;
    (void)hipRuntimeGetVersion(&ver);
    std::printf("mode %d, hipRuntimeGetVersion %d\n", mode, ver);
    int* d = nullptr;
    CK(hipMalloc(&d, sizeof(int)));
    CK(hipMemset(d, 0, sizeof(int)));
    hipStream_t O, A, B;
    CK(hipStreamCreateWithFlags(&O, hipStreamNonBlocking));
    if (mode >= 4) {  // the library's side streams of rounds 2-4: lowest priority
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, lo));
        CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, lo));
    } else {
        CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    }
    if (mode >= 3) {  // the library's shape: one eager (uncaptured) pass over the same streams and EVENTS first, as its warm-up call does
        hipEvent_t w[5];
        for (hipEvent_t& e : w) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        (void)w;
    }
    hipEvent_t eFork, eA1, eB1, eJoinA, eJoinB;
    for (hipEvent_t* e : {&eFork, &eA1, &eB1, &eJoinA, &eJoinB}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    auto body = [&]() -> int {
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, O, d);
    CK(hipEventRecord(eFork, O));
    CK(hipStreamWaitEvent(A, eFork, 0));
    CK(hipStreamWaitEvent(B, eFork, 0));
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, A, d);
    CK(hipEventRecord(eA1, A));
    if (mode >= 1) CK(hipStreamWaitEvent(B, eA1, 0));  // B waits on A's event
    if (mode >= 5) CK(hipStreamWaitEvent(B, eFork, 0));  // (mode 5: side streams wait on origin events again in between, as the library's do on every octave's event)
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, B, d);
    CK(hipEventRecord(eB1, B));
    if (mode >= 5) CK(hipStreamWaitEvent(A, eFork, 0));
    if (mode >= 2) CK(hipStreamWaitEvent(A, eB1, 0));  // ... and A on B's
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, A, d);
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, B, d);
    CK(hipEventRecord(eJoinA, A));
    CK(hipEventRecord(eJoinB, B));
    CK(hipStreamWaitEvent(O, eJoinA, 0));
    CK(hipStreamWaitEvent(O, eJoinB, 0));
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, O, d);
    return 0;
    };
    if (mode >= 3) {  // eager first (the events have been recorded outside a capture before they are recorded inside one)
        if (body()) return 1;
        CK(hipStreamSynchronize(O));
        CK(hipMemset(d, 0, sizeof(int)));
    }
    CK(hipStreamBeginCapture(O, hipStreamCaptureModeThreadLocal));
    if (body()) return 1;
    hipGraph_t g = nullptr;
    std::printf("calling hipStreamEndCapture\n");
    std::fflush(stdout);
    CK(hipStreamEndCapture(O, &g));
    hipGraphExec_t ge = nullptr;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, O));
    CK(hipStreamSynchronize(O));
    int h = -1;
    CK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
    std::printf("kernels run by the graph: %d (expected 6)\n", h);
    return h == 6 ? 0 : 2;
}
