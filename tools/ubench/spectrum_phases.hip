// DEVELOPER-ONLY: where a wave of k_spectrum_w64 spends its cycles.  The kernel is compiled here with PG_W64_PROFILE, which adds
// s_memtime reads at the phase boundaries of the frame loop and sums the differences per wave; the bench batch (16384 frames,
// 8192 bins) runs with synthetic input and the table is averaged over waves and frames.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPG_W64_PROFILE -Ipebblesdr_amd/csrc -Iinclude tools/ubench/spectrum_phases.hip -o tools/ubench/spectrum_phases
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "kernels_spectrum.h"
using namespace pg;

static bool g_settled = false;
static void run(int F, int G, const float2 *d_in, float *d_out, const float *d_win, const float2 *d_bt, float *d_p0, float *d_p1, long long *d_prof)
{
    SpectrumParams sp;
    sp.in_pitch = (long long)F * 2048;
    sp.n_frames = F;
    sp.frames_per_group = G;
    sp.scale = 1.0f / 2048.0f;
    sp.out_pitch = (long long)F * 8192;
    const int WG = (F + G - 1) / G;
    RawSrc rs{d_prof, 0, 0, 0.f, 0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    const int reps = g_settled ? 300 : 6;  // settled: 300 launches back to back, the last 100 timed as one region
    for (int it = 0; it < reps; it++) {
        if (!g_settled || it == reps - 100) hipEventRecord(e0);
        hipLaunchKernelGGL((k_spectrum_w64<-1>), dim3(WG), dim3(256), 0, 0, d_in, d_out, d_win, d_bt, (const float *)d_p0, d_p1, sp, rs);
        if (!g_settled) {
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
    }
    if (g_settled) {
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&best, e0, e1);
        best /= 100.f;
    }
    std::vector<long long> h((size_t)WG * 4 * 8);
    hipMemcpy(h.data(), d_prof, h.size() * 8, hipMemcpyDeviceToHost);
    double acc[8] = {0};
    for (size_t w = 0; w < (size_t)WG * 4; w++)
        for (int i = 0; i < 8; i++) acc[i] += (double)h[w * 8 + i];
    const double per = (double)WG * 4 * (G + 1);
    const char *names[6] = {"frame read + pre-twiddle", "transform", "fetch issue + dB + stage", "barrier B", "park + output", "barrier C"};
    double tot = 0;
    for (int i = 0; i < 6; i++) tot += acc[i] / per;
    printf("G %d (%d workgroups): %.3f ms per launch (instrumented); shader-clock ticks per wave and frame:\n", G, WG, best);
    for (int i = 0; i < 6; i++) printf("   %-26s %9.1f  (%4.1f %%)\n", names[i], acc[i] / per, 100.0 * acc[i] / per / tot);
    printf("   %-26s %9.1f   -> %.2f GHz if a wave is resident for the whole launch\n", "sum", tot, tot * (G + 1) / (best * 1e6));
}

int main()
{
    const int F = 16384;
    std::vector<float2> h((size_t)F * 2048);
    unsigned seed = 1;
    for (auto &v : h) {
        seed = seed * 1664525u + 1013904223u;
        v.x = (float)((seed >> 8) & 0xffff) / 65536.0f - 0.5f;
        seed = seed * 1664525u + 1013904223u;
        v.y = (float)((seed >> 8) & 0xffff) / 65536.0f - 0.5f;
    }
    std::vector<float> win(2048, 0.5f);
    std::vector<float2> bt(4 * 32);
    for (int q = 0; q < 4; q++)
        for (int m = 0; m < 32; m++) bt[q * 32 + m] = make_float2((float)cos(-6.283185307179586 * m * q / 128.0), (float)sin(-6.283185307179586 * m * q / 128.0));
    float2 *d_in, *d_bt;
    float *d_out, *d_win, *d_p0, *d_p1;
    long long *d_prof;
    hipMalloc(&d_in, h.size() * 8);
    hipMalloc(&d_out, (size_t)F * 8192 * 4);
    hipMalloc(&d_win, 2048 * 4);
    hipMalloc(&d_bt, bt.size() * 8);
    hipMalloc(&d_p0, 8192 * 4);
    hipMalloc(&d_p1, 8192 * 4);
    hipMalloc(&d_prof, (size_t)F * 4 * 8 * 8);
    hipMemset(d_p0, 0, 8192 * 4);
    hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_win, win.data(), 2048 * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_bt, bt.data(), bt.size() * 8, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; pass++) {
    g_settled = pass == 1;
    printf("---- %s ----\n", g_settled ? "settled (300 launches back to back)" : "cold (best of 6 single launches)");
    run(F, 32, d_in, d_out, d_win, d_bt, d_p0, d_p1, d_prof);
  }
    return 0;
}
