// Development probe for the cell-list ball query (ogc_amd/csrc/grid.hip compiled with OGC_GRID_PROBE): the C4 loss shape
// (16 clouds x 8192 points in a 60 x 4 x 80 box, radius 2, 64 samples), build and query timed apart with HIP events, cycle
// stamps of the build's phases and cycle sums of the query's phases.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ogc_amd/csrc -I include tools/bq_probe.hip -o /tmp/bq_probe
#define OGC_GRID_PROBE 1
#include "../ogc_amd/csrc/api.hip"
#include "../ogc_amd/csrc/grid.hip"
#include <vector>

int main(int argc, char **argv) {
    const int B = argc > 2 ? atoi(argv[2]) : 16, N = 8192, NS = 64; // (B: clouds per launch)
    const float radius = argc > 1 ? atof(argv[1]) : 2.0f;
    std::vector<float> h((size_t)B * N * 3);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xFFFFFF) / 16777216.0f; };
    const float scale[3] = {60.f, 4.f, 80.f};
    for (size_t i = 0; i < h.size(); ++i) h[i] = (rnd() - 0.5f) * scale[i % 3];
    float *xyz; int *idx;
    hipMalloc(&xyz, h.size() * 4); hipMalloc(&idx, (size_t)B * N * NS * 4);
    hipMemcpy(xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (int i = 0; i < 5; ++i) if (ogc_ball_query_grid(B, N, N, radius, NS, xyz, xyz, idx, s) != 0) { printf("unsupported\n"); return 1; }
    hipStreamSynchronize(s);
    const int stride_cells = GRID_MAX_CELLS + 1;
    const size_t bytes_hdr = (sizeof(GridHdr) * B + 255) / 256 * 256;
    const size_t bytes_cs = (sizeof(int) * (size_t)B * stride_cells + 255) / 256 * 256;
    char *ws = static_cast<char *>(ogc_workspace(s, bytes_hdr + bytes_cs + sizeof(float4) * (size_t)B * N));
    GridHdr *hdrs = reinterpret_cast<GridHdr *>(ws);
    int *cell_start = reinterpret_cast<int *>(ws + bytes_hdr);
    float4 *sorted_pts = reinterpret_cast<float4 *>(ws + bytes_hdr + bytes_cs);
    std::vector<int> ref((size_t)B * N * NS), out((size_t)B * N * NS);
    const int IT = 50;
    size_t bad_total = 0;
    for (int mode = 0; mode < 2; ++mode) { // general kernel | four lanes per centre
        setenv("OGC_BQ_CELLS", mode ? "1" : "0", 1);
        hipMemset(idx, 0xff, ref.size() * 4);
        unsigned long long zero[64] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(ogc_grid_probe), zero, sizeof(zero));
        float tb = 0;
        for (int i = 0; i < IT; ++i) {
            hipEventRecord(e0, s);
            launch_grid_build(B, N, radius, 0, stride_cells, xyz, hdrs, cell_start, sorted_pts, s);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float a; hipEventElapsedTime(&a, e0, e1);
            tb += a;
        }
        hipEventRecord(e0, s);
        for (int i = 0; i < IT; ++i) launch_grid_build(B, N, radius, 0, stride_cells, xyz, hdrs, cell_start, sorted_pts, s);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float bb; hipEventElapsedTime(&bb, e0, e1);
        printf("build back to back: %.2f us per launch\n", bb / IT * 1e3);
        hipEventRecord(e0, s);
        for (int i = 0; i < IT; ++i) ogc_ball_query_grid(B, N, N, radius, NS, xyz, xyz, idx, s);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float a; hipEventElapsedTime(&a, e0, e1);
        printf("---- %s: build alone %.2f us, operator back to back %.2f us per call\n", mode ? "four lanes per centre" : "general kernel",
               tb / IT * 1e3, a / IT * 1e3);
        unsigned long long pr[64];
        hipMemcpyFromSymbol(pr, HIP_SYMBOL(ogc_grid_probe), sizeof(pr));
        if (mode == 0) {
            const char *names[] = {"load+bbox", "barrier", "header", "cells+histogram", "scan part 1", "scan part 2 + cell_start", "scatter"};
            for (int i = 0; i < 7; ++i) printf("build %-28s %8llu cycles\n", names[i], pr[i + 1] - pr[i]);
            printf("build total %llu cycles\n", pr[7] - pr[0]);
        }
        const int per_wave = mode ? CPW : QPW;
        const double waves = (double)IT * B * ((N / per_wave + 63) / 64); // sampled: one wavefront in 64
        const char *qn[] = {"candidates + hit lists", "overflow (bitmap)", "rank sort", "emit"};
        for (int i = 0; i < 4; ++i) printf("query %-28s %10.1f cycles per wavefront\n", qn[i], (double)pr[16 + i] / waves);
        hipMemcpy(mode ? out.data() : ref.data(), idx, out.size() * 4, hipMemcpyDeviceToHost);
        if (mode) {
            size_t bad = 0;
            for (size_t i = 0; i < out.size(); ++i) bad += out[i] != ref[i];
            printf("rows differ from the general kernel's in %zu entries of %zu\n", bad, out.size());
            bad_total += bad;
        }
    }
    return bad_total != 0;
}
