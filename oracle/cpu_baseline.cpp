// cpu_baseline.cpp -- the CPU baseline SURVEY.md 8(d) / BASELINE.md 3 prescribe: the oracle's scalar fp64 restatement of
// the reference chain (pebble_oracle.c, compiled into this program with -O3 -march=native on the machine that runs it),
// with the frame loop in C++ and the channels / streams of a workload distributed over std::threads.
//
// TEST INFRASTRUCTURE ONLY (see pebble_oracle.h): bench.py's cpu_baseline leg builds and runs it; nothing under
// pebblesdr_amd/ or include/ may.  It is a reported baseline, not a target.
//
//   cpu_baseline <workload> <seconds> <threads|0=hardware_concurrency> [check]
//     workload 1  configs[1]: 20 Msps, one channel per thread, WFM mono + 8192-bin spectrum on every 2048-sample frame
//     workload 2  configs[2]: 2.048 Msps shared stream -> 256 USB channels (mixer + decimate + FastFIR), channels over threads
//     workload 3  configs[3] shard: 100 Msps shared stream -> 512 AM/USB channels, channels over threads
//     workload 4  configs[4] shard: 128 streams, FastFIR 2048/1025 at the stream rate + 65536-point spectrum, streams over threads
//   Every thread works through its own units frame by frame for `seconds`; the program prints ONE JSON line with the
//   1-thread and all-threads rates (unit-samples per second = IQ Msamples/s as BASELINE.json counts them).
//   `check` prints instead the first audio samples of workload 1/2's unit 0 (tests compare them with the -O2 oracle).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "pebble_oracle.h"

namespace {

using clk = std::chrono::steady_clock;

// the bench's synthetic inputs, regenerated here (formula + LCG, SURVEY.md 8d "value distributions / seeds")
struct Lcg {
    uint32_t s;
    explicit Lcg(uint32_t seed) : s(seed) {}
    double next() { s = s * 1664525u + 1013904223u; return (double)s / 4294967296.0 - 0.5; }
};

std::vector<double> make_fm(uint32_t fs, size_t n, uint32_t seed)
{
    std::vector<double> x(2 * n);
    Lcg g(seed);
    for (size_t i = 0; i < n; i++) {
        const double t = (double)i / fs;
        const double ph = 2 * M_PI * 1.0e6 * t + 75.0 * std::sin(2 * M_PI * 1000 * t);
        const double re = 0.5 * std::cos(ph) + 1e-2 * g.next(), im = 0.5 * std::sin(ph) + 1e-2 * g.next();
        x[2 * i] = std::nearbyint(re * 128) / 128.0;  // HackRF int8 shape, scale 1/128
        x[2 * i + 1] = std::nearbyint(im * 128) / 128.0;
    }
    return x;
}

std::vector<double> make_noise(size_t n, uint32_t seed, double amp)
{
    std::vector<double> x(2 * n);
    Lcg g(seed);
    for (size_t i = 0; i < 2 * n; i++) x[i] = amp * g.next();
    return x;
}

struct Work {
    int workload;
    double seconds;
    std::atomic<bool> go{false};
};

// one thread's share: units [u0, u1); returns unit-samples processed and, in *secs, the time spent processing them (the
// clock starts once the thread's units are set up: filter design is not part of the per-frame path)
uint64_t run_units(const Work &w, int u0, int u1, const std::vector<double> &shared_in, bool check, double *secs)
{
    const uint32_t nf = 2048;
    uint64_t done = 0;
    auto t_begin = clk::now();
    auto t_end = t_begin;
    auto start_clock = [&] { t_begin = clk::now(); t_end = t_begin + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(w.seconds)); };
    auto stop_clock = [&] { *secs = std::chrono::duration<double>(clk::now() - t_begin).count(); };
    if (w.workload == 4) {
        const uint32_t N = 65536;
        std::vector<po_fastfir *> ff;
        std::vector<po_spectrum *> sp;
        for (int u = u0; u < u1; u++) {
            ff.push_back(po_fastfir_new(2048, 1025));
            po_fastfir_setup(ff.back(), -50e3, 50e3, 0, 2.0e6);
            sp.push_back(po_spectrum_new(N, N, 0, 1));
        }
        std::vector<double> in = make_noise(N, 4u + (uint32_t)u0, 0.2), out(2 * (size_t)N + 4096), db(N);
        start_clock();
        do {
            for (size_t k = 0; k < ff.size(); k++) {
                for (uint32_t o = 0; o < N; o += 2048) po_fastfir_process(ff[k], 2048, in.data() + 2 * o, out.data() + 2 * o);
                po_spectrum_process(sp[k], in.data(), N, db.data());
                done += N;
            }
        } while (clk::now() < t_end);
        stop_clock();
        for (auto *f : ff) po_fastfir_free(f);
        for (auto *s : sp) po_spectrum_free(s);
        return done;
    }
    const uint32_t fs = w.workload == 1 ? 20000000u : w.workload == 2 ? 2048000u : 100000000u;
    const int C = w.workload == 2 ? 256 : 512;
    std::vector<po_receiver *> rx;
    for (int u = u0; u < u1; u++) {
        po_receiver *r = po_receiver_new(fs, nf, w.workload == 1 ? 8192 : 0, 0, 0);
        if (w.workload == 1) {
            po_receiver_set_mode(r, PO_FMM);
            po_receiver_set_mixer(r, 1.0e6);
        } else {
            const int mode = (w.workload == 3 && (u % 2 == 0)) ? PO_AM : PO_USB;
            po_receiver_set_mode(r, mode);
            po_receiver_set_mixer(r, ((double)u - C / 2.0) * (0.8 * fs / C));
            if (mode == PO_USB) po_receiver_set_filter(r, 300, 3000);
            else po_receiver_set_filter(r, -4000, 4000);
        }
        rx.push_back(r);
    }
    const size_t frames = shared_in.size() / 2 / nf;
    std::vector<double> audio(2 * (size_t)nf + 2 * 2048), db(8192);
    size_t f = 0;
    start_clock();
    do {
        const double *x = shared_in.data() + 2 * (size_t)nf * (f % frames);
        for (auto *r : rx) {
            const uint32_t na = po_receiver_process(r, x, nf, audio.data(), w.workload == 1 ? db.data() : nullptr);
            done += nf;
            if (check && na) {
                for (uint32_t i = 0; i < 16; i++) printf("%.17g %.17g\n", audio[2 * i], audio[2 * i + 1]);
                return done;
            }
        }
        f++;
    } while (clk::now() < t_end || check);
    stop_clock();
    for (auto *r : rx) po_receiver_free(r);
    return done;
}

double measure(Work &w, int threads, int units, const std::vector<double> &in)
{
    std::vector<uint64_t> done(threads, 0);
    std::vector<double> secs(threads, 0.0);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) {
        const int u0 = (int)((long long)units * t / threads), u1 = (int)((long long)units * (t + 1) / threads);
        th.emplace_back([&, t, u0, u1] { done[t] = run_units(w, u0, u1, in, false, &secs[t]); });
    }
    for (auto &t : th) t.join();
    // threads run concurrently for (about) the same span; the aggregate rate is the sum of the per-thread rates
    double rate = 0;
    for (int t = 0; t < threads; t++)
        if (secs[t] > 0) rate += (double)done[t] / secs[t];
    return rate / 1e6;
}

std::string cpu_model()
{
    FILE *f = fopen("/proc/cpuinfo", "r");
    char line[512];
    std::string m = "unknown";
    while (f && fgets(line, sizeof line, f))
        if (!strncmp(line, "model name", 10)) {
            const char *c = strchr(line, ':');
            if (c) { m = c + 2; while (!m.empty() && (m.back() == '\n' || m.back() == ' ')) m.pop_back(); }
            break;
        }
    if (f) fclose(f);
    return m;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: cpu_baseline <workload 1..4> <seconds> <threads|0> [check]\n"); return 2; }
    Work w;
    w.workload = atoi(argv[1]);
    w.seconds = atof(argv[2]);
    int threads = atoi(argv[3]);
    const bool check = argc > 4 && !strcmp(argv[4], "check");
    const int hw = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = hw > 0 ? hw : 1;
    if (w.workload < 1 || w.workload > 4) return 2;
    // one super-frame of shared input (workload 1: 64 frames; 2: 32 frames; 3: a bounded 64-frame sample of the 2048-frame super-frame)
    std::vector<double> in;
    if (w.workload == 1) in = make_fm(20000000u, 64 * 2048, 99);
    else if (w.workload == 2) in = make_noise(32 * 2048, 2, 0.1);
    else if (w.workload == 3) in = make_noise(64 * 2048, 3, 0.1);
    if (check) { double s = 0; run_units(w, 0, 1, in, true, &s); return 0; }
    // units a full run distributes: workload 1 has one channel per GPU, so every thread gets its own (weak, like the ranks);
    // the banks have a fixed unit count
    const int units_all = w.workload == 1 ? threads : w.workload == 2 ? 256 : w.workload == 3 ? 512 : 128;
    const int units_one = w.workload == 1 ? 1 : (units_all / threads > 0 ? units_all / threads : 1);  // one thread's share of the same bank
    const double one = measure(w, 1, units_one, in);
    const double all = threads > 1 ? measure(w, threads, units_all, in) : one;
    printf("{\"workload\": %d, \"threads\": %d, \"hardware_concurrency\": %d, \"cpu_model\": \"%s\", \"msamples_per_s_1thread\": %.3f, "
           "\"msamples_per_s_all_threads\": %.3f, \"seconds_per_leg\": %.1f, \"units\": %d, \"dtype\": \"f64\"}\n",
           w.workload, threads, hw, cpu_model().c_str(), one, all, w.seconds, units_all);
    return 0;
}
