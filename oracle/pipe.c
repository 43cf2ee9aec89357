/*
 * pipe.c -- oracle whole-chain driver (TEST INFRASTRUCTURE ONLY).
 *
 * Runs the reference pipe of examples/dmr-decoder.sh:19-23 / ysf-decoder.sh:19-23
 * (rrc_filter | gfsk_demodulator | dmr_decoder) for many independent channels,
 * one channel per worker thread at a time.  Used by tests as the checker and by
 * bench.py's cpu_baseline leg ("port").
 */
#include "dh_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const orc_chain_cfg* cfg;
    const float* in; size_t n_channels, stride, n;
    float* filtered;
    uint8_t* syms; size_t sym_stride; uint32_t* sym_count;
    uint8_t* out; size_t out_stride; uint32_t* out_count;
    orc_event* ev; size_t ev_stride; uint32_t* ev_count;
    size_t next; pthread_mutex_t lock;
    int error;
} job;

static int run_channel(job* j, size_t ch) {
    const orc_chain_cfg* cfg = j->cfg;
    const float* x = j->in + ch * j->stride;
    float* f = NULL; int own_f = 0;
    if (cfg->rrc) {
        if (j->filtered) f = j->filtered + ch * j->stride;
        else { f = (float*) malloc(sizeof(float) * (j->n ? j->n : 1)); own_f = 1; }
        orc_rrc* r = orc_rrc_new(cfg->rrc == 2);
        orc_rrc_process(r, x, f, j->n);
        orc_rrc_free(r);
        x = f;
    }
    int rc = 0;
    if (cfg->levels) {
        uint8_t* s = j->syms + ch * j->sym_stride;
        size_t ns = 0;
        orc_demod* d = orc_demod_new(cfg->sps, cfg->levels, cfg->invert);
        orc_demod_process(d, x, j->n, s, j->sym_stride, &ns);
        orc_demod_free(d);
        if (j->sym_count) j->sym_count[ch] = (uint32_t) ns;
        if (cfg->proto) {
            orc_decoder* dec = cfg->proto == 1 ? orc_dmr_new() : cfg->proto == 2 ? orc_ysf_new() : cfg->proto == 3 ? orc_nxdn_new() : cfg->proto == 4 ? orc_pocsag_new() : orc_dstar_new();
            if (cfg->proto == 1) orc_dmr_set_slot_filter(dec, (uint8_t) cfg->slot_filter);
            size_t no = 0, ne = 0;
            orc_decoder_process(dec, s, ns,
                                j->out ? j->out + ch * j->out_stride : NULL, j->out ? j->out_stride : 0, &no,
                                j->ev ? j->ev + ch * j->ev_stride : NULL, j->ev ? j->ev_stride : 0, &ne);
            orc_decoder_free(dec);
            if (j->out_count) j->out_count[ch] = (uint32_t) no;
            if (j->ev_count) j->ev_count[ch] = (uint32_t) ne;
        }
    }
    if (own_f) free(f);
    return rc;
}

static void* worker(void* arg) {
    job* j = (job*) arg;
    for (;;) {
        pthread_mutex_lock(&j->lock);
        size_t ch = j->next++;
        pthread_mutex_unlock(&j->lock);
        if (ch >= j->n_channels) break;
        if (run_channel(j, ch)) j->error = 1;
    }
    return NULL;
}

int orc_chain_run(const orc_chain_cfg* cfg, const float* in, size_t n_channels, size_t stride, size_t n,
                  float* filtered,
                  uint8_t* syms, size_t sym_stride, uint32_t* sym_count,
                  uint8_t* out, size_t out_stride, uint32_t* out_count,
                  orc_event* ev, size_t ev_stride, uint32_t* ev_count,
                  int n_threads) {
    job j;
    memset(&j, 0, sizeof(j));
    j.cfg = cfg; j.in = in; j.n_channels = n_channels; j.stride = stride; j.n = n;
    j.filtered = filtered;
    j.syms = syms; j.sym_stride = sym_stride; j.sym_count = sym_count;
    j.out = out; j.out_stride = out_stride; j.out_count = out_count;
    j.ev = ev; j.ev_stride = ev_stride; j.ev_count = ev_count;
    if (cfg->levels && !syms) return -1;
    pthread_mutex_init(&j.lock, NULL);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, worker, &j);
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&j.lock);
    return j.error ? -2 : 0;
}
