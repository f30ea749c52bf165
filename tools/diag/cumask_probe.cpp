// Which XCDs / CUs does a stream created with hipExtStreamCreateWithCUMask run on, for a given bit pattern -- launched directly and replayed
// from a captured hipGraph?  (dev tool, run on the GPU box; VERDICT r03 #2a: lanes confined to disjoint CU sets)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/diag/cumask_probe.cpp -o /tmp/cumask_probe && /tmp/cumask_probe
// Every workgroup records HW_REG_XCC_ID and HW_REG_HW_ID (CU / SE fields) and spins ~20 us so that the launch spreads over all CUs it may use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void probe(unsigned* out, long spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char* name, const std::vector<unsigned>& mask, bool graph) {
    hipStream_t st;
    if (mask.empty()) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    else CK(hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data()));
    const int nb = 2048;
    unsigned* d;
    CK(hipMalloc(&d, nb * 8)); CK(hipMemset(d, 0xff, nb * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long spin = 2000;   // 100 MHz wall clock: 20 us
    probe<<<nb, 64, 0, st>>>(d, spin); CK(hipStreamSynchronize(st));
    float ms = 0;
    if (graph) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        probe<<<nb, 64, 0, st>>>(d, spin);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipMemsetAsync(d, 0xff, nb * 8, st));
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    } else {
        CK(hipEventRecord(e0, st)); probe<<<nb, 64, 0, st>>>(d, spin); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h(2 * nb);
    CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> cus;   // xcc -> distinct (se, cu) ids
    for (int b = 0; b < nb; b++) {
        const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
        cus[xcc].insert(((hw >> 13) & 0x7) * 16 + ((hw >> 8) & 0xf));   // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    }
    int total = 0;
    printf("%-34s %s  %.3f ms  XCC:#CUs", name, graph ? "graph " : "direct", ms);
    for (auto& kv : cus) { printf(" %u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  (total %d)\n", total);
    CK(hipFree(d)); CK(hipStreamDestroy(st));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    auto bits = [&](auto pred) { std::vector<unsigned> m(8, 0); for (int b = 0; b < 256; b++) if (pred(b)) m[b / 32] |= 1u << (b % 32); return m; };
    for (int g = 0; g < 2; g++) {
        run("no mask", {}, g);
        run("bits 0..63 (contiguous)", bits([](int b) { return b < 64; }), g);
        run("bits 64..127 (contiguous)", bits([](int b) { return b >= 64 && b < 128; }), g);
        run("b % 8 in {0,1}", bits([](int b) { return b % 8 < 2; }), g);
        run("b % 8 in {2,3}", bits([](int b) { return b % 8 == 2 || b % 8 == 3; }), g);
        run("b % 8 == 5", bits([](int b) { return b % 8 == 5; }), g);
        run("b / 32 == 3", bits([](int b) { return b / 32 == 3; }), g);
        run("b % 4 == 0", bits([](int b) { return b % 4 == 0; }), g);
    }
    return 0;
}
