// graph_probe.hip — does a HIP graph get a small batch's ~15 dependent kernel launches past the runtime's submission rate?  (round 6, DESIGN §5:
// 1 k-read calls from ten threads are bound by ~175 k runtime submissions per second process-wide.)  T threads, a stream each; per "call": K tiny
// dependent kernels + a stream synchronisation, (a) launched one by one, (b) as one instantiated graph launched with hipGraphLaunch, (c) captured
// again every call and pushed into the instantiated graph with hipGraphExecUpdate (what a pipeline whose arguments change per call would need).
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/graph_probe tools/graph_probe.hip && tools/bin/graph_probe [T] [K] [seconds]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void k_tiny(unsigned* p, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v; }
using Clock = std::chrono::steady_clock;
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 10, K = argc > 2 ? atoi(argv[2]) : 15;
    const double secs = argc > 3 ? atof(argv[3]) : 1.5;
    for (int mode = 0; mode < 3; ++mode) {
        std::atomic<long> calls{0};
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
            hipStream_t st; hipStreamCreate(&st);
            unsigned* d; hipMalloc((void**)&d, 64); hipMemsetAsync(d, 0, 64, st);
            hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
            auto capture = [&](unsigned v) { hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal); for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_tiny, dim3(4), dim3(64), 0, st, d, v); hipStreamEndCapture(st, &g); };
            if (mode) { capture(1); hipGraphInstantiate(&ex, g, nullptr, nullptr, 0); }
            unsigned v = 1;
            while (!stop.load(std::memory_order_relaxed)) {
                if (mode == 0) for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_tiny, dim3(4), dim3(64), 0, st, d, v);
                else if (mode == 1) hipGraphLaunch(ex, st);
                else { hipGraphDestroy(g); capture(++v); hipGraphNode_t bad_node; hipGraphExecUpdateResult res; if (hipGraphExecUpdate(ex, g, &bad_node, &res) != hipSuccess) { fprintf(stderr, "update failed\n"); break; } hipGraphLaunch(ex, st); }
                hipStreamSynchronize(st);
                calls.fetch_add(1);
            }
            (void)t;
        });
        const auto t0 = Clock::now();
        std::this_thread::sleep_for(std::chrono::duration<double>(secs));
        stop = true;
        for (auto& x : th) x.join();
        const double el = std::chrono::duration<double>(Clock::now() - t0).count();
        printf("%s: T=%d K=%d: %.0f calls/s, %.1f us per call and thread, %.0f kernel submissions/s\n",
               mode == 0 ? "launch by launch" : mode == 1 ? "graph launch     " : "capture+update+launch", T, K, calls / el, 1e6 * el * T / calls, calls * (double)K / el);
    }
    return 0;
}
