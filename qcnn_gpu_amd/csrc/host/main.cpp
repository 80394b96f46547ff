// QCNN_GPU: the reference's driver program (inference/kernel.cu:74-138) for Linux / MI355X.
//
//   QCNN_GPU <ori.yuv> <anchor_prefix> <height> <width>
//
// Same positional arguments, stdout lines, log.txt record and recon_psnr.data double as the
// reference.  What the reference fixes at compile time is read from the environment:
//   QCNN_MODEL   printf pattern of the model file, %d = QP
//                (default "qvrcnn_nchw_vect_c_8bit_qfp_%d.data"; inference/kernel.cu:7)
//   QCNN_QPS     comma list of QPs (default "22": the loop bounds of inference/kernel.cu:122)
//   QCNN_FRAMES  frames per sequence (default 1 = FRAME, inference/mat.cuh:23)
//   QCNN_GPUS    number of GPUs to shard frames over (default 1; frame i -> GPU i mod G)
//   QCNN_GPU_MAP comma list of device numbers: worker g runs on device map[g mod len] instead of device g,
//                and QCNN_GPUS is then not limited to the devices present (e.g. QCNN_GPUS=2
//                QCNN_GPU_MAP=0,0 runs the two-worker sharding on one GPU)
//   QCNN_PIPELINE=1    run the timed region through qcnn_forward_blu_clip: the same H2D / forward_blu /
//                      D2H per frame, as a three-stream pipeline over groups of frames, and the PSNR
//                      sums on the device
//   QCNN_STREAM=1      the pipelined frame loop on a clip that never sits in host memory whole: reader threads
//                      (QCNN_STREAM_READERS, default 6) fill chunks of QCNN_STREAM_CHUNK frames (default: the
//                      engine's frames per launch, at most 16 MiB) by positioned reads, QCNN_STREAM_ENGINES (default 2)
//                      workers per GPU, each with an engine of its own, run qcnn_forward_blu_clip on whole chunks, writer threads (QCNN_STREAM_WRITERS, default 4) write
//                      the reconstruction, all at once; host buffers = (workers + readers + writers) chunks x 3.
//                      The time reported then INCLUDES reading and writing the files (the reference reads the
//                      whole clip before its clock starts); a breakdown goes to stderr.
//   QCNN_REPORT_RSS=1  print the peak resident set of the run on stderr when it ends
//   QCNN_PSNR_PF=1     also print "PSNR of Frame %d:%f" per frame (inference/yuv_data.cpp:109)
//   QCNN_SAVE_RECON=f  write the reconstruction as YUV 4:2:0 (inference/yuv_data.cpp:113-128)
//   QCNN_CONVERT=1     run convert_model() instead (inference/kernel.cu:17-26): HWCN -> NCHW_VECT_C,
//                      input pattern QCNN_MODEL_HWCN (default "qvrcnn_hwcn_8bit_qfp_%d.data")
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <cstdint>
#include <vector>

#include <time.h>

#include "qvrcnn.h"
#include "yuv_data.h"

// Peak resident set of this process image in MB: VmHWM of /proc/self/status (getrusage's ru_maxrss is carried over
// from the process that started this one, so it says nothing when the parent was large).
static double peak_rss_mb(void)
{
    double mb = 0;
    if (FILE *f = fopen("/proc/self/status", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            long kb;
            if (sscanf(line, "VmHWM: %ld kB", &kb) == 1) mb = kb / 1024.0;
        }
        fclose(f);
    }
    return mb;
}

static const char *env_or(const char *name, const char *dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? v : dflt;
}

static std::vector<int> parse_qps(const char *s)
{
    std::vector<int> out;
    std::string cur;
    for (const char *p = s;; ++p) {
        if (*p == ',' || *p == 0) {
            if (!cur.empty()) out.push_back(atoi(cur.c_str()));
            cur.clear();
            if (!*p) break;
        } else
            cur.push_back(*p);
    }
    return out;
}

// QCNN_DYNAMIC=1 runs the dynamic-quantisation pass instead (the commented-out alternative of the
// reference's driver, inference/kernel.cu:87,93: load_para + forward); the model file is then the
// dynamic format (step_w + weights + biases per layer).
// QCNN_FLOAT=1 runs the float32 configuration (the reference built with FLOAT_CONFIG, inference/kernel.cu:13-14:
// MODEL_NAME is then the NCHW float model).
static bool dynamic_mode() { return getenv("QCNN_DYNAMIC") != nullptr; }
static bool float_mode() { return getenv("QCNN_FLOAT") != nullptr; }
static void load_model(qvrcnn &net, char *model_fn)
{
    float_mode() ? net.load_float_para(model_fn) : dynamic_mode() ? net.load_para(model_fn) : net.load_static_para(model_fn);
}
static void run_frame(qvrcnn &net) { float_mode() ? net.forward_float() : dynamic_mode() ? net.forward() : net.forward_blu(); }

// Device of worker g: g itself, or entry g of QCNN_GPU_MAP (cyclic).
static std::vector<int> g_gpu_map;
static int device_of(int g) { return g_gpu_map.empty() ? g : g_gpu_map[g % g_gpu_map.size()]; }

// One worker per GPU: the frame loop of inference/kernel.cu:91-97 over frames g, g+G, g+2G, ...
// (`net` was built and loaded before the clock started, by this worker's device)
static void worker(int g, int ngpu, qvrcnn *net, vrcnn_data *d, int frame, int channel, int height, int width)
{
    check(qcnn_set_device(device_of(g)));  // the current device is a per-thread setting
    const long fs = (long)channel * height * width;
    for (int i = g; i < frame; i += ngpu) {
        net->load_data(d->input + i * fs);
        run_frame(*net);
        check(qcnn_sync(net->engine));
        check(qcnn_dev_d2h(d->recon + i * fs, net->I1.x_rec, (size_t)fs));
    }
}

// Pipelined variant of the worker: frames g, g+G, ... of the clip through qcnn_forward_blu_clip, the
// per-frame sums of squared differences against the original written to sse_before / sse_after.
static void worker_clip(int gpu, int ngpu, qvrcnn *net, vrcnn_data *d, int frame, int channel, int height, int width,
                        uint64_t *sse_before, uint64_t *sse_after)
{
    const size_t fs = (size_t)channel * height * width;
    const int mine = frame > gpu ? (frame - gpu + ngpu - 1) / ngpu : 0;
    if (mine == 0) return;
    check(qcnn_set_device(device_of(gpu)));
    std::vector<uint64_t> b(mine), a(mine);
    check(qcnn_forward_blu_clip(net->engine, d->input + gpu * fs, d->recon + gpu * fs, mine, fs * ngpu, d->ori + gpu * fs,
                                b.data(), a.data()));
    for (int k = 0; k < mine; ++k) {
        sse_before[gpu + (size_t)k * ngpu] = b[k];
        sse_after[gpu + (size_t)k * ngpu] = a[k];
    }
}

static void emit_record(char *input_fn, int frame, int height, int width, double psnr1, double psnr2, long long us);

// PSNR of inference/kernel.cu:105-106 (per frame too when asked), then the record
static void report(vrcnn_data &d, char *input_fn, int frame, int height, int width, long long us,
                   const std::vector<uint64_t> *sse_before = nullptr, const std::vector<uint64_t> *sse_after = nullptr,
                   bool chroma = false)
{
    if (chroma) {
        // chroma extension: stdout only; log.txt and recon_psnr.data keep the reference's luma records
        if (getenv("QCNN_PSNR_PF")) d.psnr_pf();
        printf("chroma before net:PSNR=%.3f\nchroma after quantized net:PSNR=%.3f\nchroma time:%lldus\n", d.psnr(d.input),
               d.psnr(d.recon), us);
        if (getenv("QCNN_SAVE_RECON")) d.patch_chroma_into(getenv("QCNN_SAVE_RECON"));
        return;
    }
    double psnr1, psnr2;
    if (sse_before && sse_after) {
        // the integer sums came from the device; the same doubles as vrcnn_data::psnr / psnr_pf
        const long hw = (long)height * width;
        unsigned long long tb = 0, ta = 0;
        for (int i = 0; i < frame; ++i) {
            if (getenv("QCNN_PSNR_PF")) printf("PSNR of Frame %d:%f\n", i, vrcnn_data::psnr_from_sse((*sse_after)[i], hw));
            tb += (*sse_before)[i];
            ta += (*sse_after)[i];
        }
        psnr1 = vrcnn_data::psnr_from_sse(tb, d.nSize);
        psnr2 = vrcnn_data::psnr_from_sse(ta, d.nSize);
    } else {
        if (getenv("QCNN_PSNR_PF")) d.psnr_pf();
        psnr1 = d.psnr(d.input);
        psnr2 = d.psnr(d.recon);
    }
    emit_record(input_fn, frame, height, width, psnr1, psnr2, us);
    if (getenv("QCNN_SAVE_RECON")) d.save_recon_as(getenv("QCNN_SAVE_RECON"));
}

// stdout line, log.txt record and recon_psnr.data double of inference/kernel.cu:107-115
static void emit_record(char *input_fn, int frame, int height, int width, double psnr1, double psnr2, long long us)
{
    time_t now = time(0);
    printf("\nbefore net:PSNR=%.3f\nafter quantized net:PSNR=%.3f\ntime:%lldus\n", psnr1, psnr2, us);
    FILE *logfile = fopen("log.txt", "a+");
    if (!logfile)
        printf("write file failed\n");
    else {
        fprintf(logfile,
                "\nQVRCNN test date:%sdata:%s\nframes:%d\nheight:%d\nwidth:%d\nbefore net:PSNR=%f\nafter quantized "
                "net:PSNR=%f\ntime:%lldus\n",
                ctime(&now), input_fn, frame, height, width, psnr1, psnr2, us);
        fclose(logfile);
    }
    logfile = fopen("recon_psnr.data", "ab+");
    if (!logfile)
        printf("open psnr file failed\n");
    else {
        fwrite(&psnr2, sizeof(double), 1, logfile);
        fclose(logfile);
    }
}

// QCNN_STREAM=1: the timed frame loop of inference/kernel.cu:89-101 as a pipeline over CHUNKS of a clip that is
// read, processed and written at the same time, so that host memory holds a few chunks instead of the clip
// (vrcnn_data holds frame*h*w bytes three times: 3.1 GB for 500 frames of 1920x1080):
//   readers   chunk c is read by reader c mod R with positioned reads (vrcnn_stream::read_frames) into slot c mod S
//   workers   QCNN_STREAM_ENGINES per GPU; each takes the next chunk in order as soon as it is ready and runs it through
//             qcnn_forward_blu_clip (H2D / fused kernels / D2H overlapped on three streams, integer SSE against
//             the original on the device): whole chunks per GPU, whichever GPU is free first
//   writers   chunk c is written by writer c mod W (reconstruction luma at its file offset), which frees the slot
// A slot is reused once its previous chunk has been written: S = workers + readers + writers slots of three pinned
// buffers.  The per-frame sums stay in two small arrays, so the PSNR lines are printed in frame order at the end,
// where the whole-clip path prints them.
static void testqvrcnn_stream(char *ori_fn, char *input_fn, char *model_fn, int frame, int channel, int height, int width,
                              int ngpu)
{
    using clk = std::chrono::steady_clock;
    const size_t fs = (size_t)channel * height * width;
    if (ngpu < 1) ngpu = 1;
    // Two engines (and workers) per GPU by default: a chunk goes through qcnn_forward_blu_clip as one blocking call, H2D
    // -> kernels -> D2H in a row, so a second engine on the same device is what lets the copies of one chunk run under
    // the kernels of another.
    int per_gpu = atoi(env_or("QCNN_STREAM_ENGINES", "2"));
    per_gpu = per_gpu < 1 ? 1 : per_gpu > 4 ? 4 : per_gpu;
    const int nworkers = ngpu * per_gpu;
    std::vector<qvrcnn *> nets;
    for (int k = 0; k < nworkers; ++k) {
        nets.push_back(new qvrcnn(device_of(k % ngpu), 1, channel, height, width));
        nets.back()->load_static_para(model_fn);
    }
    // frames per chunk: what one launch of the fused kernels covers, but at most 16 MiB of frames, so that large frames
    // still make enough chunks for the three stages to overlap (and the host buffers stay well below 1 GB)
    int chunk = atoi(env_or("QCNN_STREAM_CHUNK", "0"));
    if (chunk < 1) {
        check(qcnn_batch_group(nets[0]->engine, 0, &chunk));
        const long long fit = (16ll << 20) / (long long)fs;
        if (chunk > fit) chunk = fit < 1 ? 1 : (int)fit;
    }
    if (chunk > frame) chunk = frame;
    const int nchunks = (frame + chunk - 1) / chunk;
    auto threads_of = [&](const char *name, const char *dflt) {
        const int n = atoi(env_or(name, dflt));
        return n < 1 ? 1 : n > 16 ? 16 : n > nchunks ? nchunks : n;
    };
    const int R = threads_of("QCNN_STREAM_READERS", "6"), W = threads_of("QCNN_STREAM_WRITERS", "4");
    const int S = nworkers + R + W < nchunks ? nworkers + R + W : nchunks;
    for (int k = 0; k < nworkers; ++k) check(qcnn_clip_reserve(nets[k]->engine, chunk, 1));
    // slots: page-locked when the runtime grants it, ordinary memory otherwise
    std::vector<datatype *> buf(3 * (size_t)S, nullptr);
    bool pinned = true;
    for (size_t i = 0; i < buf.size() && pinned; ++i) {
        void *p = nullptr;
        pinned = qcnn_host_alloc_pinned(&p, (size_t)chunk * fs) == 0;
        buf[i] = static_cast<datatype *>(p);
    }
    if (!pinned)
        for (size_t i = 0; i < buf.size(); ++i) {
            if (buf[i]) qcnn_host_free_pinned(buf[i]);
            buf[i] = new datatype[(size_t)chunk * fs];
        }
    auto slot_ori = [&](int c) { return buf[3 * (size_t)(c % S)]; };
    auto slot_in = [&](int c) { return buf[3 * (size_t)(c % S) + 1]; };
    auto slot_rec = [&](int c) { return buf[3 * (size_t)(c % S) + 2]; };
    auto first_of = [&](int c) { return c * chunk; };
    auto count_of = [&](int c) { return frame - c * chunk < chunk ? frame - c * chunk : chunk; };

    vrcnn_stream io(ori_fn, input_fn, frame, height, width);
    if (getenv("QCNN_SAVE_RECON")) io.open_recon(getenv("QCNN_SAVE_RECON"));
    std::vector<uint64_t> sse_before(frame), sse_after(frame);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> ready(nchunks, 0), computed(nchunks, 0), written(nchunks, 0);
    std::atomic<int> next_chunk{0};
    std::atomic<long long> read_us{0}, gpu_us{0}, write_us{0};
    auto us_since = [](clk::time_point t0) { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count(); };
    auto wait_for = [&](std::vector<char> &flags, int c) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return flags[c] != 0; });
    };
    auto raise = [&](std::vector<char> &flags, int c) {
        {
            std::lock_guard<std::mutex> lk(mu);
            flags[c] = 1;
        }
        cv.notify_all();
    };

    const auto t1 = clk::now();
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r)
        th.emplace_back([&, r] {
            for (int c = r; c < nchunks; c += R) {
                if (c >= S) wait_for(written, c - S);  // the slot's previous chunk has left it
                const auto t0 = clk::now();
                io.read_frames(first_of(c), count_of(c), slot_ori(c), slot_in(c));
                read_us += us_since(t0);
                raise(ready, c);
            }
        });
    for (int g = 0; g < nworkers; ++g)
        th.emplace_back([&, g] {
            check(qcnn_set_device(device_of(g % ngpu)));  // the current device is a per-thread setting
            for (;;) {
                const int c = next_chunk++;
                if (c >= nchunks) break;
                wait_for(ready, c);
                const auto t0 = clk::now();
                check(qcnn_forward_blu_clip(nets[g]->engine, slot_in(c), slot_rec(c), count_of(c), fs, slot_ori(c),
                                            sse_before.data() + first_of(c), sse_after.data() + first_of(c)));
                gpu_us += us_since(t0);
                raise(computed, c);
            }
        });
    for (int k = 0; k < W; ++k)
        th.emplace_back([&, k] {
            for (int c = k; c < nchunks; c += W) {
                wait_for(computed, c);
                const auto t0 = clk::now();
                io.write_frames(first_of(c), count_of(c), slot_rec(c));
                write_us += us_since(t0);
                raise(written, c);
            }
        });
    for (auto &t : th) t.join();
    const long long us = us_since(t1);

    const long hw = (long)height * width;
    unsigned long long tb = 0, ta = 0;
    for (int i = 0; i < frame; ++i) {
        if (getenv("QCNN_PSNR_PF")) printf("PSNR of Frame %d:%f\n", i, vrcnn_data::psnr_from_sse(sse_after[i], hw));  // inference/yuv_data.cpp:109
        tb += sse_before[i], ta += sse_after[i];
    }
    emit_record(input_fn, frame, height, width, vrcnn_data::psnr_from_sse(tb, io.nSize), vrcnn_data::psnr_from_sse(ta, io.nSize), us);
    fprintf(stderr,
            "stream: %d frames in %d chunks of %d, %d slot(s) x 3 buffers = %.1f MB of %s host memory, %d reader(s), %d GPU worker(s) on %d GPU(s), "
            "%d writer(s); busy: read %.1f ms, gpu %.1f ms, write %.1f ms; wall %.1f ms (file time included); max RSS %.1f MB\n",
            frame, nchunks, chunk, S, 3.0 * S * chunk * fs / 1e6, pinned ? "page-locked" : "pageable", R, nworkers, ngpu, W, read_us / 1e3,
            gpu_us / 1e3, write_us / 1e3, us / 1e3, peak_rss_mb());
    for (datatype *p : buf) {
        if (pinned) qcnn_host_free_pinned(p);
        else delete[] p;
    }
    for (qvrcnn *n : nets) delete n;
}

// chroma = true: the same loop over the U and V planes (half size, two per frame) of the two files
static void testqvrcnn(char *ori_fn, char *input_fn, char *model_fn, int frame, int channel, int height, int width,
                       int ngpu, bool chroma = false)
{
    if (getenv("QCNN_STREAM") && !dynamic_mode() && !float_mode() && !chroma) {
        testqvrcnn_stream(ori_fn, input_fn, model_fn, frame, channel, height, width, ngpu);
        return;
    }
    vrcnn_data test_data(frame, height, width);
    auto read = [&]() {
        if (chroma) test_data.read_chroma(ori_fn, input_fn);
        else test_data.read_data(ori_fn, input_fn);
    };
    long long us;
    if (getenv("QCNN_PIPELINE") && !dynamic_mode() && !float_mode() && !chroma) {
        // engines and models are set up before the clock starts (inference/kernel.cu:86-90); the timed
        // region is the pipelined frame loop on every GPU, frames sharded round-robin
        if (ngpu < 1) ngpu = 1;
        std::vector<qvrcnn *> nets;
        for (int g = 0; g < ngpu; ++g) {
            nets.push_back(new qvrcnn(device_of(g), 1, channel, height, width));
            nets.back()->load_static_para(model_fn);
            check(qcnn_clip_reserve(nets.back()->engine, frame > g ? (frame - g + ngpu - 1) / ngpu : 0, 1));
        }
        test_data.read_data(ori_fn, input_fn);
        std::vector<uint64_t> sse_before(frame), sse_after(frame);
        // Part of the set-up: reading the files left the devices idle for seconds, and their shader clock needs some
        // milliseconds of load to come back.  Every engine runs the first frames of its share once, untimed (the
        // timed pass writes the same reconstruction again); QCNN_NO_WARMUP=1 leaves it out.
        if (!getenv("QCNN_NO_WARMUP")) {
            const int warm = frame < 64 * ngpu ? frame : 64 * ngpu;
            std::vector<std::thread> wt;
            for (int g = 0; g < ngpu; ++g)
                wt.emplace_back(worker_clip, g, ngpu, nets[g], &test_data, warm, channel, height, width, sse_before.data(),
                                sse_after.data());
            for (auto &t : wt) t.join();
        }
        const auto t1 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int g = 0; g < ngpu; ++g)
            th.emplace_back(worker_clip, g, ngpu, nets[g], &test_data, frame, channel, height, width, sse_before.data(),
                            sse_after.data());
        for (auto &t : th) t.join();
        us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count();
        for (qvrcnn *n : nets) delete n;
        report(test_data, input_fn, frame, height, width, us, &sse_before, &sse_after);
        return;
    }
    if (ngpu <= 1) {
        // set-up and model load happen before the clock starts, as in inference/kernel.cu:86-90;
        // the engine is kept across QPs (models switch by reloading, no rebuild)
        static qvrcnn *cached = nullptr;
        static int ch = 0, cw = 0;
        if (!cached || ch != height || cw != width) {
            delete cached;
            cached = new qvrcnn(0, 1, channel, height, width);
            ch = height;
            cw = width;
        }
        load_model(*cached, model_fn);
        read();
        const long fs = (long)channel * height * width;
        const auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < frame; i++) {
            cached->load_data(test_data.input + i * fs);
            run_frame(*cached);
            check(qcnn_sync(cached->engine));
            check(qcnn_dev_d2h(test_data.recon + i * fs, cached->I1.x_rec, (size_t)fs));
        }
        us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count();
    } else {
        // whole frames sharded round-robin, one host thread and one engine per GPU, no communication
        // between GPUs; engines and models are set up before the clock starts (inference/kernel.cu:86-90)
        std::vector<qvrcnn *> nets;
        for (int g = 0; g < ngpu; ++g) {
            nets.push_back(new qvrcnn(device_of(g), 1, channel, height, width));
            load_model(*nets.back(), model_fn);
        }
        read();
        const auto t1 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int g = 0; g < ngpu; ++g)
            th.emplace_back(worker, g, ngpu, nets[g], &test_data, frame, channel, height, width);
        for (auto &t : th) t.join();
        us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count();
        for (qvrcnn *n : nets) delete n;
    }
    report(test_data, input_fn, frame, height, width, us, nullptr, nullptr, chroma);
}

static int run_all(char *oriname, char *inputname, int height, int width)
{
    const std::vector<int> qps = parse_qps(env_or("QCNN_QPS", "22"));
    const char *pattern = env_or("QCNN_MODEL", "qvrcnn_nchw_vect_c_8bit_qfp_%d.data");
    const int frames = atoi(env_or("QCNN_FRAMES", "1"));
    int ngpu = atoi(env_or("QCNN_GPUS", "1"));
    int have = 0;
    check(qcnn_device_count(&have));
    g_gpu_map = parse_qps(env_or("QCNN_GPU_MAP", ""));
    for (int dev : g_gpu_map)
        if (dev < 0 || dev >= have) {
            printf("QCNN_GPU_MAP names device %d, %d present\n", dev, have);
            return 1;
        }
    if (g_gpu_map.empty() && ngpu > have) ngpu = have;
    if (ngpu > 64) ngpu = 64;
    char input_fn[1024], model_fn[1024];
    for (int qp : qps) {
        snprintf(input_fn, sizeof input_fn, "%sQ%d.yuv", inputname, qp);  // inference/kernel.cu:124
        snprintf(model_fn, sizeof model_fn, pattern, qp);                 // inference/kernel.cu:125
        testqvrcnn(oriname, input_fn, model_fn, frames > 0 ? frames : 1, CHANNEL, height, width, ngpu);
        check(qcnn_dev_sync());
        if (getenv("QCNN_CHROMA")) {
            // chroma extension: the same network over the half-size U and V planes (4:2:0)
            testqvrcnn(oriname, input_fn, model_fn, 2 * (frames > 0 ? frames : 1), CHANNEL, height / 2, width / 2, ngpu, true);
            check(qcnn_dev_sync());
        }
    }
    if (getenv("QCNN_REPORT_RSS")) {  // peak resident set of the whole run, on stderr
        fprintf(stderr, "max RSS %.1f MB\n", peak_rss_mb());
    }
    return 0;
}

static void convert_model(void)
{
    const std::vector<int> qps = parse_qps(env_or("QCNN_QPS", "32"));  // inference/kernel.cu:20
    for (int qp : qps)
        model_qfp_HWCN2NCHW_VECT_C(env_or("QCNN_MODEL_HWCN", "qvrcnn_hwcn_8bit_qfp_%d.data"),
                                   env_or("QCNN_MODEL", "qvrcnn_nchw_vect_c_8bit_qfp_%d.data"), qp);
}

int main(int argc, char **argv)
{
    if (getenv("QCNN_CONVERT")) {
        convert_model();
        return 0;
    }
    if (argc < 5) {
        printf("usage: %s <ori.yuv> <anchor_prefix> <height> <width>\n", argv[0]);
        return 1;
    }
    return run_all(argv[1], argv[2], atoi(argv[3]), atoi(argv[4]));
}
