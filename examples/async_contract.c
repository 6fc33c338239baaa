/* async_contract.c -- a plain-C host against the C ABI (no Python wrapper in between): pebblegpu_receiver_process only queues
 * work; reading the audio through pebblegpu_memcpy_d2h right after it must give the same bytes as reading after an explicit
 * pebblegpu_receiver_synchronize, and refilling the input through pebblegpu_memcpy_h2d right after a call must not disturb
 * that call (include/pebblegpu.h, ASYNCHRONY).  Prints "ok" or the first difference.
 * Build: gcc -O2 -Iinclude examples/async_contract.c -Lpebblesdr_amd -lpebblegpu -Wl,-rpath,$PWD/pebblesdr_amd -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pebblegpu.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, pebblegpu_last_error()); return 1; } } while (0)

int main(void)
{
    pebblegpu_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.sample_rate = 2048000.0;
    cfg.frames_per_buffer = 2048;
    cfg.n_channels = 64;
    cfg.shared_input = 1;
    cfg.max_superframes = 8;
    pebblegpu_receiver *rx = NULL;
    CHECK(pebblegpu_receiver_create(&cfg, &rx));
    pebblegpu_info info;
    CHECK(pebblegpu_receiver_info(rx, &info));
    for (uint32_t c = 0; c < cfg.n_channels; c++) {
        CHECK(pebblegpu_set_demod_mode(rx, c, PEBBLEGPU_DM_USB));
        CHECK(pebblegpu_set_mixer_freq(rx, c, -500e3 + 15e3 * c));
        CHECK(pebblegpu_set_bandpass(rx, c, 300, 3000));
    }
    const uint64_t n = 8 * info.superframe;
    float *x = (float *)malloc(sizeof(float) * 2 * n), *z = (float *)calloc(2 * n, sizeof(float));
    unsigned s = 12345u;
    for (uint64_t i = 0; i < 2 * n; i++) { s = s * 1664525u + 1013904223u; x[i] = ((float)s / 4294967296.0f - 0.5f) * 0.2f; }
    void *d_in = NULL;
    CHECK(pebblegpu_malloc(0, sizeof(float) * 2 * n, &d_in));
    uint64_t na = 0, pitch = 0;
    const size_t row = sizeof(float) * 2 * (size_t)(n / info.total_decimation);
    float *a = (float *)malloc(row), *b = (float *)malloc(row), *c2 = (float *)malloc(row);
    for (int round = 0; round < 3; round++) {
        /* reference result of this round: process, synchronize, read */
        CHECK(pebblegpu_memcpy_h2d(0, d_in, x, sizeof(float) * 2 * n));
        CHECK(pebblegpu_receiver_process(rx, d_in, n));
        /* (1) read at once, no synchronize: the copy itself must wait for the queued call */
        const void *d_audio = pebblegpu_receiver_audio(rx, &na, &pitch);
        CHECK(pebblegpu_memcpy_d2h(0, a, (const char *)d_audio + sizeof(float) * 2 * pitch * 63, row));
        CHECK(pebblegpu_receiver_synchronize(rx));
        CHECK(pebblegpu_memcpy_d2h(0, b, (const char *)d_audio + sizeof(float) * 2 * pitch * 63, row));
        if (memcmp(a, b, row)) { printf("round %d: unsynchronised read differs\n", round); return 1; }
        /* (2) overwrite the input right behind a queued call: the upload must wait, the call must still see x */
        CHECK(pebblegpu_memcpy_h2d(0, d_in, x, sizeof(float) * 2 * n));
        CHECK(pebblegpu_receiver_process(rx, d_in, n));
        CHECK(pebblegpu_memcpy_h2d(0, d_in, z, sizeof(float) * 2 * n));
        CHECK(pebblegpu_memcpy_d2h(0, c2, (const char *)d_audio + sizeof(float) * 2 * pitch * 63, row));
        double e = 0, p = 0;
        for (size_t i = 0; i < row / sizeof(float); i++) p += (double)c2[i] * c2[i];
        (void)e;
        if (!(p > 0)) { printf("round %d: the refill overtook the call (all-zero audio)\n", round); return 1; }
    }
    if (na != n / info.total_decimation) { printf("unexpected audio length\n"); return 1; }
    /* (3) the library's pinned double buffer: batches of 16-bit pairs filled, submitted and processed through the two slots in turn
     * must give the bytes a twin receiver gives that is handed the same pairs from a device buffer through
     * pebblegpu_receiver_process_raw, batch by batch, and a slot cannot be submitted again while its call is in flight */
    {
        const uint64_t raw_bytes = 4 * n;  /* int16 I, Q */
        pebblegpu_receiver *twin = NULL;
        CHECK(pebblegpu_receiver_create(&cfg, &twin));
        for (uint32_t c = 0; c < cfg.n_channels; c++) {
            CHECK(pebblegpu_set_demod_mode(twin, c, PEBBLEGPU_DM_USB));
            CHECK(pebblegpu_set_mixer_freq(twin, c, -500e3 + 15e3 * c));
            CHECK(pebblegpu_set_bandpass(twin, c, 300, 3000));
        }
        pebblegpu_receiver *fresh = NULL;  /* the receiver above has history: the ingest side starts from a new one as well */
        CHECK(pebblegpu_receiver_create(&cfg, &fresh));
        for (uint32_t c = 0; c < cfg.n_channels; c++) {
            CHECK(pebblegpu_set_demod_mode(fresh, c, PEBBLEGPU_DM_USB));
            CHECK(pebblegpu_set_mixer_freq(fresh, c, -500e3 + 15e3 * c));
            CHECK(pebblegpu_set_bandpass(fresh, c, 300, 3000));
        }
        void *d_raw = NULL;
        CHECK(pebblegpu_malloc(0, raw_bytes, &d_raw));
        short *batch_raw = (short *)malloc(raw_bytes), *h = NULL;
        float *want = (float *)malloc(row), *got = (float *)malloc(row);
        for (int batch = 0; batch < 5; batch++) {
            const uint32_t slot = (uint32_t)(batch & 1);
            for (uint64_t i = 0; i < 2 * n; i++) { s = s * 1664525u + 1013904223u; batch_raw[i] = (short)((int)(s >> 16) - 32768) / 8; }
            CHECK(pebblegpu_receiver_ingest_acquire(fresh, slot, raw_bytes, (void **)&h));
            memcpy(h, batch_raw, raw_bytes);
            CHECK(pebblegpu_receiver_ingest_submit(fresh, slot, raw_bytes));
            CHECK(pebblegpu_receiver_process_ingested(fresh, slot, PEBBLEGPU_IQ_S16, 0, 1.0, n));
            if (pebblegpu_receiver_ingest_submit(fresh, slot, raw_bytes) == 0) { printf("a slot in flight was submitted again\n"); return 1; }
            const void *ag = pebblegpu_receiver_audio(fresh, &na, &pitch);
            CHECK(pebblegpu_memcpy_d2h(0, got, (const char *)ag + sizeof(float) * 2 * pitch * 63, row));
            CHECK(pebblegpu_memcpy_h2d(0, d_raw, batch_raw, raw_bytes));
            CHECK(pebblegpu_receiver_process_raw(twin, PEBBLEGPU_IQ_S16, 0, 1.0, d_raw, n));
            const void *aw = pebblegpu_receiver_audio(twin, &na, &pitch);
            CHECK(pebblegpu_memcpy_d2h(0, want, (const char *)aw + sizeof(float) * 2 * pitch * 63, row));
            double p = 0;
            for (size_t i = 0; i < row / sizeof(float); i++) p += (double)want[i] * want[i];
            if (!(p > 0) || memcmp(got, want, row) != 0) { printf("batch %d through the pinned slots differs from the device-buffer batch\n", batch); return 1; }
        }
        if (pebblegpu_receiver_process_ingested(fresh, 0, PEBBLEGPU_IQ_F32, 0, 1.0, n) == 0) { printf("a slot too small for the format was accepted\n"); return 1; }
        CHECK(pebblegpu_receiver_destroy(twin));
        CHECK(pebblegpu_receiver_destroy(fresh));
        free(batch_raw); free(want); free(got);
        CHECK(pebblegpu_free(0, d_raw));
    }
    CHECK(pebblegpu_free(0, d_in));
    CHECK(pebblegpu_receiver_destroy(rx));
    printf("ok\n");
    return 0;
}
