// rds.hip -- RdsCore: the RDS branch of Demod_WFM::processDataStereo (application/demod/demod_wfm.cpp:296-357, 488-786) on the device
// (kernels_rds.h) and the host-side replay of m_RdsGroupQueue with its consumer (Demod::fmStereo, application/demod.cpp:196-226).
#include <algorithm>
#include "kernels_rds.h"
#include "receiver.h"

namespace pg {

static long long cdiv_ll(long long a, long long b) { return (a + b - 1) / b; }

static int upload(double **d, const std::vector<double> &h)
{
    PG_HIP(hipMalloc((void **)d, sizeof(double) * h.size()));
    PG_HIP(hipMemcpy(*d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    return 0;
}

int RdsCore::init(uint32_t channels, double demod_rate, long long max_n)
{
    C = channels;
    cap = max_n;
    des = design::rds_design(demod_rate);
    const size_t nst = des.stages.size();
    D = 1 << nst;
    memset(&pp, 0, sizeof(pp));
    pp.osc_turns = des.osc_turns;
    pp.nco_lo = des.nco_lo; pp.nco_hi = des.nco_hi; pp.alpha = des.alpha; pp.beta = des.beta;
    pp.b0 = des.bitsync.b0; pp.b2 = des.bitsync.b2; pp.a1 = des.bitsync.a1; pp.a2 = des.bitsync.a2;
    pp.mtaps = (int)des.matched.size();
    pp.log_cap = 4096;
    if (int rc = raw.alloc((int)C, 64, max_n)) return rc;
    long long len = max_n;
    for (size_t j = 0; j <= nst; j++) {
        std::vector<double> h;
        int nw = 0;
        if (j < nst) {
            h = design::downconvert_stage_response(des.stages[j]);
            nw = des.stages[j] == 0 ? 1 : 0;  // the CIC3 ends on the pair's odd sample (downconvert.cpp:517-533)
        } else {
            h.assign(des.lp.rbegin(), des.lp.rend());
        }
        if ((int)h.size() > kMaxTaps) return fail(PEBBLEGPU_E_UNSUPPORTED, "RDS stage of %zu taps", h.size());
        double *t = nullptr;
        if (int rc = upload(&t, h)) return rc;
        d_taps.push_back(t);
        ntaps.push_back((int)h.size());
        newest.push_back(nw);
        HistBuf b;
        if (int rc = b.alloc((int)C, 2 * ((int)h.size() + 1), 2 * (len + 2))) return rc;  // rows of double2 in 8-byte units
        st.push_back(b);
        if (j < nst) len /= 2;
    }
    const long long out_cap = len + 2;
    if (int rc = mag.alloc((int)C, pp.mtaps + 2, out_cap)) return rc;
    PG_HIP(hipMalloc((void **)&d_lp, sizeof(double2) * (size_t)out_cap * C));
    PG_HIP(hipMalloc((void **)&d_data, sizeof(double) * (size_t)out_cap * C));
    PG_HIP(hipMemset(d_data, 0, sizeof(double) * (size_t)out_cap * C));
    PG_HIP(hipMalloc((void **)&d_state, sizeof(RdsState) * C));
    PG_HIP(hipMemset(d_state, 0, sizeof(RdsState) * C));  // initRds, :524-537 (and m_RdsLastData, which the reference never sets)
    PG_HIP(hipMalloc((void **)&d_log, sizeof(RdsEvent) * (size_t)pp.log_cap * C));
    PG_HIP(hipMemset(d_log, 0, sizeof(RdsEvent) * (size_t)pp.log_cap * C));
    std::vector<double> amp(kRdsAmpTab);
    design::oscillator_amplitudes(amp.data(), kRdsAmpTab);
    if (int rc = upload(&d_amp, amp)) return rc;
    if (int rc = upload(&d_matched, des.matched)) return rc;
    host.assign(C, Host());
    on = true;
    return 0;
}

void RdsCore::release()
{
    raw.release(); mag.release();
    for (auto &b : st) b.release();
    st.clear();
    void *p[] = {d_lp, d_data, d_state, d_log, d_amp, d_matched, d_lptaps};
    for (void *q : p) if (q) (void)hipFree(q);
    for (double *t : d_taps) if (t) (void)hipFree(t);
    d_taps.clear(); ntaps.clear(); newest.clear();
    d_lp = nullptr; d_data = nullptr; d_state = nullptr; d_log = nullptr; d_amp = d_matched = d_lptaps = nullptr;
    on = false;
}

int RdsCore::check(long long n, int block) const
{
    if (!on) return 0;
    const size_t nst = des.stages.size();
    if (block <= 0) block = (int)n;
    if (n > cap) return fail(PEBBLEGPU_E_SIZE, "%lld samples exceed this object's capacity", n);
    if (n % D != 0 || block % D != 0)
        return fail(PEBBLEGPU_E_SIZE, "dmFMS: calls and frames must be multiples of %d samples (the RDS down-converter's %zu decimate-by-2 stages)", D, nst);
    for (size_t j = 0; j < nst; j++)
        if ((block >> j) < ntaps[j])  // CHalfBandDecimateBy2's "safety net" (downconvert.cpp:361-362) would drop samples unfiltered
            return fail(PEBBLEGPU_E_SIZE, "dmFMS: a frame of %d samples is shorter than the RDS down-converter's stage %zu needs", block, j);
    return 0;
}

int RdsCore::run(hipStream_t s, const float2 *in, long long in_pitch, long long n, const double *d_hilb, const int *d_list, int n_list, int block)
{
    if (!on || n_list == 0) return 0;
    const size_t nst = des.stages.size();
    if (int rc = check(n, block)) return rc;
    if (block <= 0) block = (int)n;
    const dim3 blk(256);
    launch(k_rds_discrim, dim3((unsigned)cdiv_ll(n, 256), n_list), blk, s, in, in_pitch, n, (const RdsState *)d_state, reinterpret_cast<double *>(raw.data()), raw.pitch,
           d_list);
    launch(k_rds_hilbert_mix, dim3((unsigned)cdiv_ll(n, 256), n_list), blk, s, reinterpret_cast<const double *>(raw.data()), raw.pitch, n, d_hilb,
           (const double *)d_amp, (const RdsState *)d_state, pp.osc_turns, reinterpret_cast<double2 *>(st[0].data()), st[0].pitch / 2, d_list);
    std::vector<TailJob> jobs;
    jobs.push_back(TailJob{raw.data(), raw.pitch, n, raw.hist, 0, nullptr, 0});
    long long len = n;
    for (size_t j = 0; j <= nst; j++) {
        const bool last = j == nst;
        const long long n_out = last ? len : len / 2;
        double2 *dst = last ? d_lp : reinterpret_cast<double2 *>(st[j + 1].data());
        const long long dst_pitch = last ? (long long)(cap / D + 2) : st[j + 1].pitch / 2;
        launch(k_rds_fir, dim3((unsigned)cdiv_ll(n_out, 256), n_list), blk, s, reinterpret_cast<const double2 *>(st[j].data()), st[j].pitch / 2, dst, dst_pitch, n_out,
               last ? 1 : 2, newest[j], (const double *)d_taps[j], ntaps[j], d_list);
        jobs.push_back(TailJob{st[j].data(), st[j].pitch, 2 * len, st[j].hist, 0, nullptr, 0});
        len = n_out;
    }
    RdsParams q = pp;
    q.block = block / D;
    launch(k_rds_pll, dim3((unsigned)cdiv_ll(n_list, 64)), dim3(64), s, (const double2 *)d_lp, (long long)(cap / D + 2), len, q, d_state,
           reinterpret_cast<double *>(mag.data()), mag.pitch, d_list, n_list);
    launch(k_rds_matched, dim3((unsigned)cdiv_ll(len, 256), n_list), blk, s, reinterpret_cast<const double *>(mag.data()), mag.pitch, len, (const double *)d_matched,
           pp.mtaps, d_data, (long long)(cap / D + 2), d_list);
    launch(k_rds_bits, dim3((unsigned)cdiv_ll(n_list, 64)), dim3(64), s, (const double *)d_data, (long long)(cap / D + 2), len, q, d_state, d_log, in, in_pitch, n,
           d_list, n_list);
    jobs.push_back(TailJob{mag.data(), mag.pitch, len, mag.hist, 0, nullptr, 0});
    if (int rc = run_save_tails(s, jobs, C)) return rc;
    PG_HIP(hipGetLastError());
    last_len = len;
    return 0;
}

int RdsCore::collect(hipStream_t s, uint32_t ch)
{
    if (!on) return 0;
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    RdsState stt;
    PG_HIP(hipMemcpyAsync(&stt, d_state + ch, sizeof(stt), hipMemcpyDeviceToHost, s));
    PG_HIP(hipStreamSynchronize(s));
    Host &h = host[ch];
    const unsigned long long total = stt.n_events;
    unsigned long long from = h.seen;
    if (total - from > (unsigned long long)pp.log_cap) {  // (hours of groups between two reads)
        h.lost += total - from - (unsigned long long)pp.log_cap;
        from = total - (unsigned long long)pp.log_cap;
    }
    std::vector<RdsEvent> ev((size_t)(total - from));
    if (!ev.empty()) {
        const unsigned long long cap_l = (unsigned long long)pp.log_cap;
        const unsigned long long a = from % cap_l, cnt = total - from, first = std::min(cnt, cap_l - a);
        const RdsEvent *base = d_log + (size_t)ch * pp.log_cap;
        PG_HIP(hipMemcpy(ev.data(), base + a, sizeof(RdsEvent) * first, hipMemcpyDeviceToHost));
        if (cnt > first) PG_HIP(hipMemcpy(ev.data() + first, base, sizeof(RdsEvent) * (cnt - first), hipMemcpyDeviceToHost));
    }
    // getNextRdsGroupData, demod_wfm.cpp:763-786, once per frame for `frames` frames (it does nothing on an empty queue)
    auto pops = [&h](long long frames) {
        while (frames > 0 && h.head != h.tail) {
            const RdsGroup g = h.q[h.tail++];
            if (h.tail >= 100) h.tail = 0;
            const bool diff = g.a != h.last.a || g.b != h.last.b || g.c != h.last.c || g.d != h.last.d;
            if (diff) h.last = g;
            h.out.push_back(g);
            h.changed.push_back(diff ? 1 : 0);
            frames--;
        }
    };
    for (const RdsEvent &e : ev) {
        if (e.frame > h.frames) {  // the frames in front of this entry's have ended: their pops come first
            pops(e.frame - h.frames);
            h.frames = e.frame;
        }
        if (e.flags & 1u) h.head = h.tail = 0;  // :642: the queue is cleared, then the zero group goes in
        h.q[h.head++] = RdsGroup{e.a, e.b, e.c, e.d};
        if (h.head >= 100 && !(e.flags & 1u)) h.head = 0;  // (the clear path does not wrap: head is 1 there)
    }
    if (stt.frames > h.frames) {
        pops(stt.frames - h.frames);
        h.frames = stt.frames;
    }
    h.seen = total;
    return 0;
}

int RdsCore::groups(hipStream_t s, uint32_t ch, RdsGroup *g, unsigned char *changed, uint32_t cap_out, uint32_t *n_out)
{
    if (n_out) *n_out = 0;
    if (!on) return 0;
    if (int rc = collect(s, ch)) return rc;
    Host &h = host[ch];
    const uint32_t n = (uint32_t)std::min<size_t>(h.out.size(), cap_out);
    for (uint32_t i = 0; i < n; i++) {
        if (g) g[i] = h.out[i];
        if (changed) changed[i] = h.changed[i];
    }
    h.out.erase(h.out.begin(), h.out.begin() + n);
    h.changed.erase(h.changed.begin(), h.changed.begin() + n);
    if (n_out) *n_out = n;
    return 0;
}

int RdsCore::signal(hipStream_t s, uint32_t ch, double *data, uint32_t cap_out, uint32_t *n_out)
{
    if (n_out) *n_out = 0;
    if (!on) return 0;
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    const uint32_t n = (uint32_t)std::min<long long>(last_len, cap_out);
    PG_HIP(hipStreamSynchronize(s));
    if (n && data) PG_HIP(hipMemcpy(data, d_data + (size_t)ch * (size_t)(cap / D + 2), sizeof(double) * n, hipMemcpyDeviceToHost));
    if (n_out) *n_out = (uint32_t)last_len;
    return 0;
}

}  // namespace pg
