// C-ABI: fused streaming session - encode -> fp16 CLS -> sliding-window head without the CLS rows leaving
// HBM ("cbas_fused_run").  The reference does this as two threads with a file in
// between (EncodeThread writes _cls.h5, ClassificationThread reads it back: backend/workthreads.py:316-328,
// :488-498; cbas.py:423-440 chunk loop, :497-551 window loop); the numerical contract is unchanged because
// the head still consumes the CLS rows AFTER their round-to-fp16 (what the file would have held).
//
// Built only on the public entry points of the encoder and the head (no access to their internals): batches
// alternate over the encoder's slots / compute lanes, a segment of frames is classified as soon as its
// right-hand context (seq_len/2 rows) has been encoded by batches the session's stream is already ordered
// after, in groups of `classify_every` frames, so the head never drains the encoder lanes mid-clip.
#include <atomic>
#include <chrono>
#include <new>
#include <thread>

#include "api_common.h"

struct cbas_fused {
    cbas_enc* enc;
    cbas_head* head;
    int device;
    int D, C, half, max_batch;
    int64_t capacity, classify_every;
    float temperature;
    uint16_t* cls16 = nullptr;      // [capacity][D] IEEE half, device
    float* probs = nullptr;         // [capacity][C], device
    hipStream_t st = nullptr;       // the head runs here; encoder batches are chained to it by events.  It is the ENCODER'S COPY
                                    // STREAM, not a stream of the session's own: the head's waits already order the host->HBM
                                    // copies (a copy into a slot waits for what was queued here), and one active stream fewer
                                    // keeps the process at three busy hardware queues - lanes 0 / 1 and this one - which is
                                    // worth 6 % on the pinned-host path (cbas_enc_copy_stream)
    bool own_stream = false;
    int64_t encoded = 0, classified = 0, landed = 0;
    struct Busy { int64_t n = 0; uint64_t seq = 0; } busy[CBAS_ENC_SLOTS];
    int next_slot = 0;
    uint64_t seq = 0;
    hipEvent_t clip_done = nullptr; // recorded on st when a clip's last operation (tail classify, copy-out) has been queued
    bool clip_pending = false;
    // rows copied out WHILE the clip runs (cbas_fused_stream_rows): every STREAM_STEP landed rows one device->host copy on st
    // with an event; a single consumer thread (the file writer) follows them through cbas_fused_rows_ready
    static constexpr int NT = 128;
    static constexpr int64_t STREAM_STEP = 512;
    struct Ticket { hipEvent_t ev = nullptr; int64_t upto = 0; } tickets[NT];
    std::atomic<uint64_t> n_tickets{0}, consumed{0};
    std::atomic<int64_t> ready_rows{0};
    uint16_t* stream_host = nullptr;
    int64_t copied = 0;
};

namespace {

// order the session's stream after the oldest batches still in flight, oldest first
int drain(cbas_fused* f) {
    for (;;) {
        int best = -1;
        for (int s = 0; s < CBAS_ENC_SLOTS; ++s)
            if (f->busy[s].n && (best < 0 || f->busy[s].seq < f->busy[best].seq)) best = s;
        if (best < 0) return CBAS_OK;
        int rc = cbas_enc_wait_stream(f->enc, best, f->st);
        if (rc) return rc;
        f->landed += f->busy[best].n;
        f->busy[best].n = 0;
    }
}

// queue the device->host copy of the landed rows [copied, upto) and publish its ticket
int queue_row_copy(cbas_fused* f, int64_t upto) {
    if (!f->stream_host || upto <= f->copied) return CBAS_OK;
    const uint64_t n = f->n_tickets.load(std::memory_order_relaxed);
    if (n - f->consumed.load(std::memory_order_acquire) >= (uint64_t)cbas_fused::NT) return CBAS_OK;   // consumer far behind: later
    cbas_fused::Ticket& t = f->tickets[n % cbas_fused::NT];
    if (!t.ev) HIP_TRY(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(f->stream_host + f->copied * f->D, f->cls16 + f->copied * f->D, (size_t)(upto - f->copied) * f->D * 2,
                           hipMemcpyDeviceToHost, f->st));
    HIP_TRY(hipEventRecord(t.ev, f->st));
    t.upto = upto;
    f->copied = upto;
    f->n_tickets.store(n + 1, std::memory_order_release);
    return CBAS_OK;
}

int classify(cbas_fused* f, int64_t count, int64_t n_rows) {
    if (!f->head) { f->classified += count; return CBAS_OK; }       // encode-only session
    int rc = cbas_head_infer_f16_range(f->head, f->cls16, n_rows, f->classified, count, f->temperature,
                                       f->probs + f->classified * f->C, nullptr, f->st);
    if (rc) return rc;
    f->classified += count;
    return CBAS_OK;
}

int push(cbas_fused* f, const uint8_t* frames, bool host, int n, int height, int width, int64_t frame_stride,
         int64_t row_stride, int64_t pixel_stride, void* after_stream) {
    if (!f) return cbas_fail(CBAS_EINVAL, "null session");
    if (n <= 0) return cbas_fail(CBAS_EINVAL, "n=%d", n);
    if (f->encoded + n > f->capacity)
        return cbas_fail(CBAS_EINVAL, "session capacity %lld exceeded (%lld encoded + %d)", (long long)f->capacity,
                         (long long)f->encoded, n);
    HIP_TRY(hipSetDevice(f->device));
    for (int i = 0; i < n; i += f->max_batch) {
        const int m = n - i < f->max_batch ? n - i : f->max_batch;
        const int slot = f->next_slot;
        f->next_slot = (f->next_slot + 1) % CBAS_ENC_SLOTS;
        if (f->busy[slot].n) {                           // slots are recycled in submission order
            int rc = cbas_enc_wait_stream(f->enc, slot, f->st);
            if (rc) return rc;
            f->landed += f->busy[slot].n;
            f->busy[slot].n = 0;
        }
        uint16_t* rows = f->cls16 + f->encoded * f->D;
        const uint8_t* src = frames + (int64_t)i * frame_stride;
        int rc = host ? cbas_enc_submit_u8_host_dev(f->enc, slot, src, m, height, width, frame_stride, row_stride,
                                                    pixel_stride, nullptr, rows, f->st)
                      : cbas_enc_submit_u8(f->enc, slot, src, m, height, width, frame_stride, row_stride, pixel_stride,
                                           nullptr, rows, after_stream);
        if (rc) return rc;
        f->busy[slot].n = m;
        f->busy[slot].seq = ++f->seq;
        f->encoded += m;
    }
    if (f->stream_host && f->landed - f->copied >= cbas_fused::STREAM_STEP) {
        int rc = queue_row_copy(f, f->landed);
        if (rc) return rc;
    }
    const int64_t ready = f->landed - f->half - f->classified;       // frames with their full right context
    if (ready >= f->classify_every) return classify(f, ready, f->landed);
    return CBAS_OK;
}

}  // namespace

extern "C" int cbas_fused_create(cbas_enc* enc, cbas_head* head, int64_t capacity_frames, float temperature,
                                 int64_t classify_every, cbas_fused** out) {
    if (!enc || !out) return cbas_fail(CBAS_EINVAL, "null argument");
    *out = nullptr;
    cbas_enc_config ec;
    cbas_head_config hc{};
    int rc = cbas_enc_get_config(enc, &ec);
    if (rc) return rc;
    if (head) {
        rc = cbas_head_get_config(head, &hc);
        if (rc) return rc;
    } else {                                   // encode-only session: the chunk loop of encode_file with the rows kept in HBM
        hc.in_features = ec.hidden_size; hc.out_features = 0; hc.seq_len = 1;
    }
    if (hc.in_features != ec.hidden_size)
        return cbas_fail(CBAS_EINVAL, "head expects %d features, the encoder emits %d", hc.in_features, ec.hidden_size);
    if (capacity_frames <= 0) return cbas_fail(CBAS_EINVAL, "capacity_frames=%lld", (long long)capacity_frames);
    cbas_fused* f = new (std::nothrow) cbas_fused();
    if (!f) return cbas_fail(CBAS_ENOMEM, "out of host memory");
    f->enc = enc; f->head = head;
    f->D = ec.hidden_size; f->C = hc.out_features; f->half = hc.seq_len / 2; f->max_batch = ec.max_batch;
    f->capacity = capacity_frames; f->temperature = temperature;
    f->classify_every = classify_every > 0 ? classify_every : 1024;
    hipError_t e = hipGetDevice(&f->device);
    if (e == hipSuccess) e = hipMalloc(&f->cls16, (size_t)capacity_frames * f->D * sizeof(uint16_t));
    if (e == hipSuccess && f->C > 0) e = hipMalloc(&f->probs, (size_t)capacity_frames * f->C * sizeof(float));
    if (e == hipSuccess) {
        f->st = (hipStream_t)cbas_enc_copy_stream(enc);
        if (!f->st) { e = hipStreamCreateWithFlags(&f->st, hipStreamNonBlocking); f->own_stream = true; }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->clip_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        cbas_fail(e == hipErrorOutOfMemory ? CBAS_ENOMEM : CBAS_EHIP, "cbas_fused_create: %s", hipGetErrorString(e));
        cbas_fused_destroy(f);
        return e == hipErrorOutOfMemory ? CBAS_ENOMEM : CBAS_EHIP;
    }
    *out = f;
    return CBAS_OK;
}

extern "C" void cbas_fused_destroy(cbas_fused* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    (void)drain(f);
    if (f->st) { (void)hipStreamSynchronize(f->st); if (f->own_stream) (void)hipStreamDestroy(f->st); }
    if (f->clip_done) (void)hipEventDestroy(f->clip_done);
    for (auto& t : f->tickets)
        if (t.ev) (void)hipEventDestroy(t.ev);
    if (f->cls16) (void)hipFree(f->cls16);
    if (f->probs) (void)hipFree(f->probs);
    delete f;
}

extern "C" int cbas_fused_reset(cbas_fused* f) {
    if (!f) return cbas_fail(CBAS_EINVAL, "null session");
    HIP_TRY(hipSetDevice(f->device));
    int rc = drain(f);
    if (rc) return rc;
    // the previous clip's rows are about to be overwritten: wait for THIS session's last clip only - the stream is shared
    // with the encoder's copies and with other sessions, whose clips may still be draining (two sessions alternate when
    // clips are pipelined back to back: cbas_amd/pipeline.py ClipRunner)
    if (f->clip_pending) { HIP_TRY(hipEventSynchronize(f->clip_done)); f->clip_pending = false; }
    else if (f->encoded > 0) HIP_TRY(hipStreamSynchronize(f->st));          // a clip that was never finished
    f->encoded = f->classified = f->landed = 0;
    f->stream_host = nullptr;                                           // (its consumer is done: the caller joined it)
    f->copied = 0;
    f->n_tickets.store(0); f->consumed.store(0); f->ready_rows.store(0);
    return CBAS_OK;
}

extern "C" int cbas_fused_stream_rows(cbas_fused* f, uint16_t* cls_f16_host) {
    if (!f) return cbas_fail(CBAS_EINVAL, "null session");
    if (f->encoded != 0 || f->n_tickets.load() != 0) return cbas_fail(CBAS_ESTATE, "cbas_fused_stream_rows: call it right after cbas_fused_reset");
    f->stream_host = cls_f16_host;
    return CBAS_OK;
}

extern "C" int64_t cbas_fused_rows_ready(cbas_fused* f, int32_t block) {
    if (!f) return -1;
    if (hipSetDevice(f->device) != hipSuccess) return -1;
    const uint64_t n = f->n_tickets.load(std::memory_order_acquire);
    uint64_t c = f->consumed.load(std::memory_order_relaxed);
    int64_t ready = f->ready_rows.load(std::memory_order_relaxed);
    bool wait = block != 0;
    while (c < n) {
        const cbas_fused::Ticket& t = f->tickets[c % cbas_fused::NT];
        if (wait) { if (hipEventSynchronize(t.ev) != hipSuccess) return -1; wait = false; }
        else if (hipEventQuery(t.ev) != hipSuccess) { (void)hipGetLastError(); break; }
        ready = t.upto;
        ++c;
    }
    f->ready_rows.store(ready, std::memory_order_relaxed);
    f->consumed.store(c, std::memory_order_release);
    return ready;
}

extern "C" int cbas_fused_push_u8_host(cbas_fused* f, const uint8_t* frames_host, int n, int height, int width,
                                       int64_t frame_stride, int64_t row_stride, int64_t pixel_stride) {
    return push(f, frames_host, true, n, height, width, frame_stride, row_stride, pixel_stride, nullptr);
}

extern "C" int cbas_fused_push_u8(cbas_fused* f, const uint8_t* frames_dev, int n, int height, int width,
                                  int64_t frame_stride, int64_t row_stride, int64_t pixel_stride, void* after_stream) {
    return push(f, frames_dev, false, n, height, width, frame_stride, row_stride, pixel_stride, after_stream);
}

namespace {
// mode 0: the host waits; 1: `stream` waits; 2: nobody waits (cbas_fused_wait does, later)
int finish_impl(cbas_fused* f, uint16_t* cls_f16_host, float* probs_host, const uint16_t** cls_f16_dev, const float** probs_dev,
                int64_t* n_frames, int mode, hipStream_t stream) {
    if (!f) return cbas_fail(CBAS_EINVAL, "null session");
    HIP_TRY(hipSetDevice(f->device));
    int rc = drain(f);
    if (rc) return rc;
    if (f->encoded > f->classified) {
        rc = classify(f, f->encoded - f->classified, f->encoded);      // the clip's right edge replicates its last row
        if (rc) return rc;
    }
    if (f->stream_host) {                                               // rows have been leaving all along: the rest
        // (a consumer that fell NT tickets behind makes queue_row_copy skip: wait for it, the clip is over anyway)
        for (int spins = 0; f->copied < f->encoded; ++spins) {
            rc = queue_row_copy(f, f->encoded);
            if (rc) return rc;
            if (f->copied < f->encoded) {
                if (spins > 20000) return cbas_fail(CBAS_ESTATE, "cbas_fused_finish: the consumer of cbas_fused_rows_ready has stalled");
                std::this_thread::sleep_for(std::chrono::microseconds(500));      // at most ~10 s
            }
        }
    } else if (cls_f16_host)
        HIP_TRY(hipMemcpyAsync(cls_f16_host, f->cls16, (size_t)f->encoded * f->D * 2, hipMemcpyDeviceToHost, f->st));
    if (probs_host && f->C > 0)
        HIP_TRY(hipMemcpyAsync(probs_host, f->probs, (size_t)f->encoded * f->C * 4, hipMemcpyDeviceToHost, f->st));
    if (cls_f16_dev) *cls_f16_dev = f->cls16;
    if (probs_dev) *probs_dev = f->probs;
    if (n_frames) *n_frames = f->encoded;
    HIP_TRY(hipEventRecord(f->clip_done, f->st));
    f->clip_pending = true;
    if (mode == 1) {                                                    // `stream` waits for the results, the host does not
        HIP_TRY(hipStreamWaitEvent(stream, f->clip_done, 0));
    } else if (mode == 0) {
        HIP_TRY(hipEventSynchronize(f->clip_done));                     // results are complete at return
        f->clip_pending = false;
        return cbas_enc_check_finite(f->enc);                           // CBAS_ERANGE instead of NaN rows in a file
    }
    return CBAS_OK;
}
}  // namespace

extern "C" int cbas_fused_finish(cbas_fused* f, uint16_t* cls_f16_host, float* probs_host, const uint16_t** cls_f16_dev,
                                 const float** probs_dev, int64_t* n_frames, void* stream) {
    return finish_impl(f, cls_f16_host, probs_host, cls_f16_dev, probs_dev, n_frames, stream ? 1 : 0, (hipStream_t)stream);
}

extern "C" int cbas_fused_finish_async(cbas_fused* f, uint16_t* cls_f16_host, float* probs_host, int64_t* n_frames) {
    return finish_impl(f, cls_f16_host, probs_host, nullptr, nullptr, n_frames, 2, nullptr);
}

extern "C" int cbas_fused_wait(cbas_fused* f) {
    if (!f) return cbas_fail(CBAS_EINVAL, "null session");
    HIP_TRY(hipSetDevice(f->device));
    if (f->clip_pending) {
        HIP_TRY(hipEventSynchronize(f->clip_done));
        f->clip_pending = false;
        return cbas_enc_check_finite(f->enc);
    }
    return CBAS_OK;
}
