/*
 * A plain-C consumer of include/cattus_hip.h that replays the call sequence of the Rust binding
 * (integration/rust/hip.rs as NNetwork::evaluate_impl drives it, engine/src/net/mod.rs:89-103):
 *
 *     cattus_hip_create  ->  T threads, each one blocking cattus_hip_apply(1 leaf) per position  ->
 *     cattus_hip_stats   ->  cattus_hip_destroy
 *
 * i.e. the reference's threading model: one search thread per game, every thread blocked in Batcher::apply
 * (engine/src/util/batch.rs:49-177), the leaves of the threads blocked together sharing a batch.
 *
 *   consumer BLOB PLANES OUT dtype batch_size threads plane_words
 *
 * BLOB: weight blob; PLANES: raw u64 words [n][planes][plane_words]; OUT: f32 [n][moves + 1] (logits, value), then the
 * stats line on stdout.  No Python, no ctypes: gcc -std=c99 consumer.c -lcattus_hip -lpthread.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cattus_hip.h"

typedef struct worker {
    cattus_eval* ev;
    const uint64_t* planes;
    float* out;
    uint32_t first, stride, n, words, moves;
    int rc;
    char err[256];
} worker;

static void* run(void* arg) {
    worker* w = (worker*)arg;
    for (uint32_t i = w->first; i < w->n; i += w->stride) {
        float* row = w->out + (size_t)i * (w->moves + 1);
        const int rc = cattus_hip_apply(w->ev, w->planes + (size_t)i * w->words, 1, row, row + w->moves);
        if (rc != CATTUS_OK) {
            w->rc = rc;
            snprintf(w->err, sizeof w->err, "%s", cattus_hip_last_error()); /* thread-local message */
            return NULL;
        }
    }
    return NULL;
}

static void* slurp(const char* path, size_t* size) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *size = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(*size ? *size : 1);
    if (p && fread(p, 1, *size, f) != *size) {
        free(p);
        p = NULL;
    }
    fclose(f);
    return p;
}

int main(int argc, char** argv) {
    if (argc != 8) {
        fprintf(stderr, "usage: %s BLOB PLANES OUT dtype batch_size threads plane_words\n", argv[0]);
        return 2;
    }
    size_t blob_size = 0, planes_size = 0;
    void* blob = slurp(argv[1], &blob_size);
    uint64_t* planes = (uint64_t*)slurp(argv[2], &planes_size);
    if (!blob || !planes) {
        fprintf(stderr, "cannot read inputs\n");
        return 2;
    }
    cattus_eval_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.device = 0;
    cfg.dtype = (uint32_t)atoi(argv[4]);
    cfg.max_batch = (uint32_t)atoi(argv[5]);
    cfg.plane_words = (uint32_t)atoi(argv[7]);
    cfg.flush_us = 200;
    const int threads = atoi(argv[6]);

    cattus_eval* ev = NULL;
    if (cattus_hip_create(blob, blob_size, &cfg, &ev) != CATTUS_OK) {
        fprintf(stderr, "create: %s\n", cattus_hip_last_error());
        return 1;
    }
    cattus_net_desc d;
    cattus_hip_desc(ev, &d);
    const uint32_t words = d.planes * cfg.plane_words;
    const uint32_t n = (uint32_t)(planes_size / (8u * words));
    float* out = (float*)calloc((size_t)n * (d.moves + 1), sizeof(float));

    pthread_t* tid = (pthread_t*)calloc((size_t)threads, sizeof *tid);
    worker* ws = (worker*)calloc((size_t)threads, sizeof *ws);
    for (int t = 0; t < threads; t++) {
        ws[t].ev = ev, ws[t].planes = planes, ws[t].out = out;
        ws[t].first = (uint32_t)t, ws[t].stride = (uint32_t)threads, ws[t].n = n, ws[t].words = words, ws[t].moves = d.moves;
        pthread_create(&tid[t], NULL, run, &ws[t]);
    }
    int failed = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(tid[t], NULL);
        if (ws[t].rc) {
            fprintf(stderr, "thread %d: %s (%d)\n", t, ws[t].err, ws[t].rc);
            failed = 1;
        }
    }
    /* a bad call is an error code and a message, never an abort: n = 0 (planes_to_tensor asserts 1 <= n, net/mod.rs:122-127) */
    float dummy;
    const int bad = cattus_hip_eval(ev, planes, 0, &dummy, &dummy);
    cattus_stats st;
    cattus_hip_stats(ev, &st);
    printf("{\"leaves\": %u, \"moves\": %u, \"batches\": %llu, \"positions\": %llu, \"full_batches\": %llu, \"run_seconds_ema\": %.9f, "
           "\"empty_eval_status\": %d}\n",
           n, d.moves, (unsigned long long)st.batches, (unsigned long long)st.positions, (unsigned long long)st.full_batches,
           st.run_seconds_ema, bad);
    cattus_hip_destroy(ev);

    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(out, sizeof(float), (size_t)n * (d.moves + 1), f) != (size_t)n * (d.moves + 1)) failed = 1;
    if (f) fclose(f);
    free(out), free(tid), free(ws), free(blob), free(planes);
    return failed;
}
