// Headless camera-stream driver of the throughput path: ONE PROCESS PER GPU, one camera stream per process
// (BASELINE configs 4 and 5).  It is to the batched path what `Harris` / `DoG` are to the reference's
// per-image main()s (Harris_corners.cpp:146-193, Diff_of_Gauss.cpp:727-877): the same detection, fed
// with batches of frames instead of one imread(), with the display replaced by one JSON line.
//
//   Stream [--mode device|hostfed] [--frames 256] [--batches 8] [--warmup 2] [--rows 1080 --cols 1920]
//          [--octaves 4] [--source synth|<file of raw 8-bit frames>] [--dump <file>] [--rdv-selftest] [--pipelines 1]
//          [--compact 0|1] [--expand N]   (host-fed: 16-byte records over PCIe; N host threads rebuild SLAM::point in the timed loop)
//          [--numa 0|1]   (default 1: threads + pinned staging buffers on the NUMA node of the rank's GPU, gpu_locality.hpp)
//          [--lists candidates|localize|orient|describe]   (how much of the DoG executable runs per frame:
//           the contrast-8 candidate list, + FeaturePointLocalization, + filterKeypoints, + SIFT descriptors)
//
//   device   frames are uploaded once and stay in HBM; every step = BatchDetector::detect_device +
//            the RCCL all-gather of the {harris, dog} counts behind it (bench.py's step; --pipelines 2 lets the
//            kernels of two consecutive batches overlap - see BatchDetector::Options::pipelines for when that pays);
//   hostfed  every batch starts in pinned host memory and its keypoint lists end there, uploads /
//            kernels / downloads of consecutive batches overlapped (BatchDetector::submit / collect);
//            the all-gather of a batch's counts follows its kernels on the same stream.
//
// Ranks: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT as torchrun sets them (or by hand:
// `for r in 0 1 ..; do RANK=$r WORLD_SIZE=N LOCAL_RANK=$r ./Stream & done`); the GPU is LOCAL_RANK, the
// camera stream id is RANK.  Counts cross ranks with ncclAllGather from librccl (count_exchange.hpp) -
// no torch, no MPI.  Timing follows the bench contract: barrier, K steps, barrier, max over ranks.
// --dump writes the keypoint lists of the last batch (hostfed: as collected; device: downloaded after the
// run) for the parity test: u32 {magic 'VSKP', n_frames, rows, cols}, then per frame u32 {n_harris, n_dog,
// harris_total, dog_total} + n_harris vslam_kp + n_dog vslam_point.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "batch_detector.hpp"
#include "count_exchange.hpp"
#include "gpu_locality.hpp"
#include "imgio.hpp"
#include "vslam_cxx.hpp"

namespace {

struct Args {
    std::string mode = "device", source = "synth", dump, lists = "candidates";
    int frames = 256, batches = 8, warmup = 6, rows = 1080, cols = 1920, octaves = 4, pipelines = 1;
    bool rdv_selftest = false, no_allgather = false, no_rccl = false, no_tuner = true;
    int hw_queues = 0;
    std::string side_priority;  // "" = by mode: host-fed low (with its 12 queues), device-resident the library's default (main)
    int compact = 1;  // host-fed: SLAM::point lists cross PCIe as 16-byte records (BatchDetector::Options::compact_points)
    // host-fed + compact: threads that rebuild every frame's 24-byte SLAM::point records from the 16-byte ones INSIDE the timed
    // loop (what a consumer of the reference's std::vector<SLAM::point> needs; 0 = the packed records are the delivered form)
    int expand = 0;
    // 1 (default): the rank's threads and pinned staging buffers go to the NUMA node of its GPU (gpu_locality.hpp); 0: left alone
    int numa = 1;
    // --numa-probe: no GPU - resolve --bdf under --sysfs, print the placement (report only unless --numa-bind), exit
    bool numa_probe = false, numa_bind = false;
    std::string sysfs = "/sys", bdf;
};

Args parse(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) throw std::runtime_error("missing value after " + k);
            return argv[++i];
        };
        if (k == "--mode") a.mode = val();
        else if (k == "--frames") a.frames = std::stoi(val());
        else if (k == "--batches") a.batches = std::stoi(val());
        else if (k == "--pipelines") a.pipelines = std::stoi(val());  // batches whose kernels may overlap (BatchDetector::Options::pipelines)
        else if (k == "--warmup") a.warmup = std::stoi(val());
        else if (k == "--rows") a.rows = std::stoi(val());
        else if (k == "--cols") a.cols = std::stoi(val());
        else if (k == "--octaves") a.octaves = std::stoi(val());
        else if (k == "--source") a.source = val();
        else if (k == "--dump") a.dump = val();
        else if (k == "--lists") a.lists = val();  // candidates (default) | localize | orient | describe: how much of the DoG executable runs per frame
        else if (k == "--compact") a.compact = std::stoi(val());  // 0: 24-byte SLAM::point records on the wire (the round-4 form)
        else if (k == "--expand") a.expand = std::stoi(val());  // N host threads re-expand every collected batch to SLAM::point inside the timed loop
        else if (k == "--numa") a.numa = std::stoi(val());
        else if (k == "--numa-probe") a.numa_probe = true;
        else if (k == "--numa-bind") a.numa_bind = true;
        else if (k == "--sysfs") a.sysfs = val();
        else if (k == "--bdf") a.bdf = val();
        else if (k == "--rdv-selftest") a.rdv_selftest = true;
        else if (k == "--no-allgather") a.no_allgather = true;  // diagnosis only: the per-step collective left out
        else if (k == "--no-tuner") a.no_tuner = true;          // (the default since round 5)
        else if (k == "--tuner") a.no_tuner = false;            // opt in to the library's comparison of three side-stream pairs (vslam_ctx_tune_side_streams)
        else if (k == "--side-priority") a.side_priority = val();  // low | main: vslam_ctx_set_side_stream_priority
        else if (k == "--hw-queues") a.hw_queues = std::stoi(val());  // GPU_MAX_HW_QUEUES for this process (set in main before HIP starts)
        else if (k == "--no-rccl") a.no_rccl = true;            // diagnosis only: single rank without a communicator
        else throw std::runtime_error("unknown argument " + k);
    }
    if ((a.lists != "candidates" && a.lists != "localize" && a.lists != "orient" && a.lists != "describe") || (a.mode != "device" && a.mode != "hostfed") || a.frames <= 0 || a.batches <= 0 || a.warmup < 0 || a.rows <= 0 || a.cols <= 0 || a.octaves < 0)
        throw std::runtime_error("bad arguments");
    return a;
}

// n frames of the camera stream `stream_id` starting at frame `first` into dst (dense rows): the synthetic
// generator of SURVEY.md section 8d (imgio::synthetic, identical to visualslam_amd/synth.py), or a file of
// raw 8-bit frames read cyclically.
void fill_frames(const Args& a, int stream_id, int first, int n, uint8_t* dst) {
    const size_t N = (size_t)a.rows * a.cols;
    if (a.source == "synth") {
        const int nt = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                for (int f = t; f < n; f += nt) {
                    const cv::Mat m = imgio::synthetic(a.rows, a.cols, first + f, stream_id);
                    std::memcpy(dst + (size_t)f * N, m.data, N);
                }
            });
        for (auto& t : th) t.join();
        return;
    }
    FILE* fp = std::fopen(a.source.c_str(), "rb");
    if (!fp) throw std::runtime_error("cannot open " + a.source);
    std::fseek(fp, 0, SEEK_END);
    const long total = std::ftell(fp) / (long)N;
    if (total <= 0) {
        std::fclose(fp);
        throw std::runtime_error(a.source + ": shorter than one frame");
    }
    for (int f = 0; f < n; ++f) {
        std::fseek(fp, (long)(((long)first + f) % total) * (long)N, SEEK_SET);
        if (std::fread(dst + (size_t)f * N, 1, N, fp) != N) {
            std::fclose(fp);
            throw std::runtime_error(a.source + ": read error");
        }
    }
    std::fclose(fp);
}

void dump_lists(const std::string& path, int rows, int cols, const vslam::BatchResult& r) {
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("cannot write " + path);
    const uint32_t head[4] = {0x504b5356u, (uint32_t)r.n_frames, (uint32_t)rows, (uint32_t)cols};
    std::fwrite(head, 4, 4, fp);
    for (int f = 0; f < r.n_frames; ++f) {
        const vslam::FrameKeypoints k = r.frame(f);
        const uint32_t h[4] = {(uint32_t)k.n_harris, (uint32_t)k.n_dog, k.harris_total, k.dog_total};
        std::fwrite(h, 4, 4, fp);
        std::fwrite(k.harris, sizeof(vslam_kp), k.n_harris, fp);
        std::fwrite(k.dog, sizeof(vslam_point), k.n_dog, fp);
    }
    std::fclose(fp);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char** argv) {
    try {
        // The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES (default 4) hardware queues per priority level,
        // and which queue a stream lands on decides whether the library's low-priority side streams run or crawl (DESIGN
        // section 5.4).  Rounds 3-4 set the variable to 12 here and opted in to the stream tuner; since round 5 the library's
        // join watchdog (include/vslam.h: vslam_ctx_join_watch_report) handles an unlucky layout by itself, so this host
        // changes nothing in the environment: --hw-queues N / --tuner are there for the comparison runs.
        // The host-fed pipeline is this HOST's own: it adds an upload, a download and a packing stream, and a copy stream that
        // shares a hardware queue with a compute stream serialises with it (host-fed 7.6-8.5 k frames/s with 1 or 3 queues
        // against 12.9 k with 12) - that mode keeps asking for 12 queues, unless the caller has set the variable.
        {
            bool hostfed_mode = false, given = false;
            for (int i = 1; i + 1 < argc; ++i) {
                if (std::string(argv[i]) == "--mode" && std::string(argv[i + 1]) == "hostfed") hostfed_mode = true;
                if (std::string(argv[i]) == "--hw-queues") ::setenv("GPU_MAX_HW_QUEUES", argv[i + 1], 1), given = true;  // before HIP starts
            }
            if (hostfed_mode && !given) ::setenv("GPU_MAX_HW_QUEUES", "12", 0);
        }
        const Args a = parse(argc, argv);
        const vslam::RankEnv env = vslam::RankEnv::from_environment();
        if (a.rdv_selftest) {  // the TCP hand-off of the RCCL id alone (no GPU): rank 0's bytes must reach every rank
            unsigned char id[128];
            for (int i = 0; i < 128; ++i) id[i] = env.rank == 0 ? (unsigned char)(i * 7 + 3) : 0;
            vslam::tcp_broadcast_from_rank0(env, id, sizeof(id));
            unsigned sum = 0;
            for (int i = 0; i < 128; ++i) sum = sum * 31 + id[i];
            std::printf("{\"exe\": \"Stream\", \"rdv_selftest\": true, \"rank\": %d, \"world\": %d, \"id_hash\": %u}\n", env.rank, env.world, sum);
            return 0;
        }
        if (a.numa_probe) {  // the sysfs side of the placement alone (no GPU): tests/test_gpu_locality_cpu.py feeds it a fake tree
            const vslam::locality::Placement pl = vslam::locality::place_near_pci(env.rank, env.local_rank, a.bdf, a.numa_bind, a.sysfs);
            const std::vector<int> now = vslam::locality::current_affinity();
            std::printf("{\"exe\": \"Stream\", \"numa_probe\": true, \"placement\": %s, \"affinity_after\": \"%s\"}\n", pl.json().c_str(),
                        vslam::locality::format_cpulist(now).c_str());
            return 0;
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device: the product has no CPU fallback");
        const int device = env.local_rank % ndev;
        // BEFORE the detector allocates its pinned staging buffers and before any worker thread exists: this thread - and so
        // every thread created from here on - moves to the CPUs of the GPU's NUMA node (gpu_locality.hpp)
        vslam::locality::Placement placement;
        placement.rank = env.rank, placement.gpu = device, placement.cpus = vslam::locality::current_affinity();
        placement.note = "--numa 0";
        if (a.numa) {
            char bdf[64] = {};
            if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) == hipSuccess)
                placement = vslam::locality::place_near_pci(env.rank, device, bdf, true, a.sysfs);
            else
                placement.note = "hipDeviceGetPCIBusId failed";
        }
        if (env.world > 1 || std::getenv("VSLAM_PRINT_PLACEMENT"))  // one line per rank on stderr (rank 0's is also in the JSON line)
            std::fprintf(stderr, "Stream: placement %s\n", placement.json().c_str());
        const size_t N = (size_t)a.rows * a.cols;
        const bool hostfed = a.mode == "hostfed";

        vslam::BatchDetector::Options opt;
        opt.device = device;
        opt.rows = a.rows, opt.cols = a.cols, opt.batch = a.frames;
        opt.host_fed = hostfed;
        opt.pipelines = a.pipelines;
        opt.compact_points = hostfed && a.compact != 0;
        opt.yielding_side_streams = a.side_priority.empty() ? (hostfed && !std::getenv("VSLAM_SIDE_PRIORITY")) : a.side_priority == "low";
        opt.tune_side_streams = !a.no_tuner;  // --tuner: the library compares three pairs of (yielding) side streams during the warm-up; off by default
        opt.localize = a.lists != "candidates", opt.orient = a.lists == "orient" || a.lists == "describe", opt.describe = a.lists == "describe";
        if (a.octaves != 4) {
            opt.custom_params = true;
            vslam_params_default(&opt.params, a.rows, a.cols);
            opt.params.n_octaves = a.octaves;
        }
        vslam::BatchDetector det(opt);
        // RCCL communicator over all ranks (collective).  --no-rccl (diagnosis, one rank only): no communicator at
        // all, the "all-gather" of the single rank is a 16-byte device copy.
        struct Exchange {
            std::unique_ptr<vslam::CountExchange> cx;
            uint64_t* d_one = nullptr;
            int rank = 0;
            void all_gather_async(const uint64_t* d_local, void* s) {
                if (cx) return cx->all_gather_async(d_local, s);
                (void)hipMemcpyAsync(d_one, d_local, 16, hipMemcpyDeviceToDevice, (hipStream_t)s);
            }
            std::vector<uint64_t> fetch(void* s) {
                if (cx) return cx->fetch(s);
                std::vector<uint64_t> v(2);
                (void)hipMemcpyAsync(v.data(), d_one, 16, hipMemcpyDeviceToHost, (hipStream_t)s);
                (void)hipStreamSynchronize((hipStream_t)s);
                return v;
            }
            void barrier(void* s) {
                if (cx) return cx->barrier(s);
                (void)hipStreamSynchronize((hipStream_t)s);
            }
            double max_over_ranks(double v, void* s) { return cx ? cx->max_over_ranks(v, s) : v; }
        } ex;
        if (a.no_rccl) {
            if (env.world != 1) throw std::runtime_error("--no-rccl is for single-rank runs");
            if (hipMalloc((void**)&ex.d_one, 16) != hipSuccess || hipMemset(ex.d_one, 0, 16) != hipSuccess) throw std::runtime_error("hipMalloc");
        } else {
            ex.cx = std::make_unique<vslam::CountExchange>(env, device, vslam::CountExchange::backend_from_environment());
        }
        // Consecutive batches run on alternating streams (pipelines), but every collective of the communicator goes to ONE
        // stream, in batch order: the exchange stream waits for a batch's totals, and the batch that reuses the pair of
        // totals waits for the collective that read it (hold_totals_until).
        hipStream_t cs = (hipStream_t)det.stream();
        if (det.pipelines() > 1 && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("hipStreamCreate");
        struct Ev {
            hipEvent_t totals = nullptr, sent = nullptr;
        };
        std::vector<Ev> ring(16);
        for (Ev& e : ring)
            if (hipEventCreateWithFlags(&e.totals, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e.sent, hipEventDisableTiming) != hipSuccess)
                throw std::runtime_error("hipEventCreate");
        size_t gathers = 0;
        auto gather_counts = [&] {
            Ev& e = ring[gathers++ % ring.size()];
            (void)hipEventRecord(e.totals, (hipStream_t)det.stream());
            (void)hipStreamWaitEvent(cs, e.totals, 0);
            ex.all_gather_async(det.device_totals(), cs);  // 16 bytes per rank behind the kernels: no host round trip
            (void)hipEventRecord(e.sent, cs);
            det.hold_totals_until(e.sent);
        };

        // the camera stream: stream_id = rank.  One batch of frames in pinned host memory.
        uint8_t* h_frames = (uint8_t*)vslam::BatchDetector::alloc_pinned((size_t)a.frames * N);
        fill_frames(a, env.rank, 0, a.frames, h_frames);

        std::vector<uint64_t> all;
        double dt = 0, steady_ms = 0;
        const vslam::BatchResult* last = nullptr;
        vslam::BatchResult dev_last;
        std::vector<uint64_t> dl_off;
        std::vector<uint32_t> dl_cnt;
        std::vector<vslam_kp> dl_h;
        std::vector<vslam_point> dl_p;
        if (!hostfed) {
            uint8_t* d_frames = nullptr;
            if (hipMalloc((void**)&d_frames, (size_t)a.frames * N) != hipSuccess) throw std::runtime_error("hipMalloc(frames)");
            if (hipMemcpy(d_frames, h_frames, (size_t)a.frames * N, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("upload");
            auto step = [&] {
                det.detect_device(d_frames, N, a.frames);
                if (!a.no_allgather) gather_counts();
            };
            for (int i = 0; i < a.warmup; ++i) step();
            det.sync();
            ex.barrier(cs);
            const double t0 = now_s();
            for (int i = 0; i < a.batches; ++i) step();
            det.sync();
            ex.barrier(cs);
            dt = now_s() - t0;
            all = ex.fetch(cs);
            if (!a.dump.empty()) {  // download the lists of the last batch for the parity check (not timed)
                const vslam_batch_out& o = det.device_outputs();
                const vslam_params& p = det.params();
                dl_off.assign(2 * (a.frames + 1), 0);
                dl_cnt.assign(2 * a.frames, 0);
                if (o.harris_counts) (void)hipMemcpy(dl_cnt.data(), o.harris_counts, 4 * a.frames, hipMemcpyDeviceToHost);
                if (o.dog_counts) (void)hipMemcpy(dl_cnt.data() + a.frames, o.dog_counts, 4 * a.frames, hipMemcpyDeviceToHost);
                for (int f = 0; f < a.frames; ++f) {
                    dl_off[f + 1] = dl_off[f] + std::min<uint32_t>(dl_cnt[f], p.harris_cap);
                    dl_off[a.frames + 1 + f + 1] = dl_off[a.frames + 1 + f] + std::min<uint32_t>(dl_cnt[a.frames + f], p.dog_cap);
                }
                dl_h.resize(dl_off[a.frames]);
                dl_p.resize(dl_off[2 * a.frames + 1]);
                for (int f = 0; f < a.frames; ++f) {
                    const size_t nh = dl_off[f + 1] - dl_off[f], np = dl_off[a.frames + 2 + f] - dl_off[a.frames + 1 + f];
                    if (nh) (void)hipMemcpy(dl_h.data() + dl_off[f], o.harris_kps + (size_t)f * p.harris_cap, nh * sizeof(vslam_kp), hipMemcpyDeviceToHost);
                    if (np) (void)hipMemcpy(dl_p.data() + dl_off[a.frames + 1 + f], o.dog_points + (size_t)f * p.dog_cap, np * sizeof(vslam_point), hipMemcpyDeviceToHost);
                }
                dev_last.n_frames = a.frames;
                dev_last.harris_offsets = dl_off.data(), dev_last.dog_offsets = dl_off.data() + a.frames + 1;
                dev_last.harris_counts = dl_cnt.data(), dev_last.dog_counts = dl_cnt.data() + a.frames;
                dev_last.harris = dl_h.data(), dev_last.dog = dl_p.data();
                dev_last.harris_records = dl_h.size(), dev_last.dog_records = dl_p.size();
                last = &dev_last;
            }
            (void)hipFree(d_frames);
        } else {
            // the all-gather of a batch's counts is enqueued right behind its kernels (device-side totals): the
            // host never waits for it inside the loop
            const int depth = std::max(1, opt.slots);
            bool truncated = false;
            auto submit = [&] {
                det.submit(h_frames, a.frames);
                if (!a.no_allgather) gather_counts();
            };
            std::vector<double> t_collect;
            // --expand N: the consumer's side of compact_points - every frame's DoG records back to SLAM::point (24 bytes), N
            // threads over the frames of the batch, into buffers that are reused (the cost of delivering the reference's record
            // form on the host; ADVICE r5: the default figure delivers the packed form and says so)
            std::vector<std::vector<vslam_point>> xbuf((size_t)std::max(0, a.expand));
            unsigned long long expanded_records = 0;
            auto expand_all = [&](const vslam::BatchResult& r) {
                if (a.expand <= 0 || !r.dog16) return;
                std::vector<std::thread> th;
                for (int t = 0; t < a.expand; ++t)
                    th.emplace_back([&, t] {
                        for (int f = t; f < r.n_frames; f += a.expand) {
                            const uint64_t d0 = std::min(r.dog_offsets[f], r.dog_records), d1 = std::min(r.dog_offsets[f + 1], r.dog_records);
                            if (xbuf[t].size() < d1 - d0) xbuf[t].resize(d1 - d0);
                            vslam_points16_expand(r.dog16 + d0, (size_t)(d1 - d0), xbuf[t].data());
                        }
                    });
                for (auto& t : th) t.join();
                expanded_records += std::min(r.dog_offsets[r.n_frames], r.dog_records);
            };
            auto run = [&](int nb) {
                int sub = 0;
                t_collect.clear();
                for (; sub < std::min(depth, nb); ++sub) submit();
                for (int k = 0; k < nb; ++k) {
                    const vslam::BatchResult& r = det.collect();
                    truncated |= r.truncated;
                    if (sub < nb) submit(), ++sub;  // the next batch is on its way before this one is consumed
                    expand_all(r);
                    t_collect.push_back(now_s());
                    last = &r;
                }
            };
            if (a.warmup) run(a.warmup);
            det.sync();
            ex.barrier(cs);
            const double t0 = now_s();
            run(a.batches);
            det.sync();
            ex.barrier(cs);
            dt = now_s() - t0;
            all = ex.fetch(cs);
            if (truncated) std::fprintf(stderr, "Stream: rank %d: lists truncated (raise the caps or the host budget)\n", env.rank);
            // steady state of the pipeline: the mean interval between collect() returns over the middle of the run (the
            // whole-run figure also carries the first upload and the last download, which nothing overlaps; with two
            // pipelines the intervals alternate, so a median would pick one of the two)
            const size_t skip = (size_t)depth + 1;
            if (t_collect.size() >= 2 * skip + 4) {
                const size_t i0 = skip, i1 = t_collect.size() - 1 - skip;
                steady_ms = (t_collect[i1] - t_collect[i0]) / (double)(i1 - i0) * 1e3;
            }
        }
        unsigned long long oriented_rank = 0;  // this rank's oriented points of the last batch (reported, not all-gathered)
        if (opt.orient) {
            if (hostfed && last && last->oriented_counts) {
                for (int f = 0; f < last->n_frames; ++f) oriented_rank += last->oriented_counts[f];
            } else if (!hostfed && det.device_outputs().oriented_counts) {
                std::vector<uint32_t> oc(a.frames);
                (void)hipMemcpy(oc.data(), det.device_outputs().oriented_counts, 4 * (size_t)a.frames, hipMemcpyDeviceToHost);
                for (uint32_t v : oc) oriented_rank += v;
            }
        }
        int probe_replaced = 0, probe_flat = 0;
        (void)vslam_ctx_side_stream_report(det.context(), &probe_replaced, &probe_flat);
        int jw_level = 0, jw_done = 0;
        float jw_lag = -1.0f;
        (void)vslam_ctx_join_watch_report(det.context(), &jw_level, &jw_done, &jw_lag);
        const char* hwq = std::getenv("GPU_MAX_HW_QUEUES");
        const double dt_max = ex.max_over_ranks(dt, cs);
        if (!a.dump.empty() && last) dump_lists(a.dump, a.rows, a.cols, *last);
        uint64_t gh = 0, gd = 0;
        for (int r = 0; r < env.world; ++r) gh += all[2 * r], gd += all[2 * r + 1];
        const bool tcp = ex.cx && ex.cx->backend() == vslam::CountExchange::Backend::Tcp;
        std::string by_rank = "[";
        for (int r = 0; r < env.world; ++r)
            by_rank += (r ? ", [" : "[") + std::to_string(all[2 * r]) + ", " + std::to_string(all[2 * r + 1]) + "]";
        by_rank += "]";
        if (env.rank == 0) {
            const double fps = (double)a.frames * a.batches * env.world / dt_max;
            std::printf("{\"exe\": \"Stream\", \"host\": \"C++ (BatchDetector) + %s\", \"mode\": \"%s\", \"n_gpus\": %d, \"frames_per_batch\": %d, "
                        "\"batches\": %d, \"warmup\": %d, \"rows\": %d, \"cols\": %d, \"octaves\": %d, \"frames_per_sec\": %.2f, \"ms_per_batch\": %.4f, "
                        "\"keypoints_per_batch\": {\"harris\": %llu, \"dog\": %llu}, \"keypoints_per_sec\": %.1f, \"rank0_counts\": [%llu, %llu], "
                        "\"steady_ms_per_batch\": %.4f, \"steady_frames_per_sec\": %.2f, \"counts_by_rank\": %s, \"lists\": \"%s\", \"oriented_points_rank0_last_batch\": %llu, \"pipelines\": %d, \"side_stream_pair\": %d, \"side_stream_tuner\": %d, "
                        "\"join_watch\": {\"level\": %d, \"done\": %d, \"last_lag_fraction\": %.4f}, \"compact_points\": %d, \"expand_threads\": %d, \"delivered_records\": \"%s\", \"gpu_max_hw_queues\": \"%s\", \"placement\": %s}\n",
                        a.no_rccl ? "no communicator (--no-rccl)" : tcp ? "TCP rehearsal exchange (VSLAM_COUNT_BACKEND=tcp)" : "RCCL ncclAllGather", a.mode.c_str(),
                        env.world, a.frames, a.batches, a.warmup, a.rows, a.cols, det.params().n_octaves, fps, dt_max / a.batches * 1e3,
                        (unsigned long long)gh, (unsigned long long)gd, (double)(gh + gd) * a.batches / dt_max, (unsigned long long)all[0],
                        (unsigned long long)all[1], steady_ms, steady_ms > 0 ? a.frames * 1e3 / steady_ms * env.world : 0.0, by_rank.c_str(), a.lists.c_str(), oriented_rank, det.pipelines(), probe_replaced, probe_flat,
                        jw_level, jw_done, (double)jw_lag, (int)(hostfed && a.compact != 0), hostfed ? a.expand : 0,
                        !hostfed ? "device lists (SLAM::point, 24 B)" : (a.compact == 0 ? "SLAM::point (24 B) over PCIe" : (a.expand > 0 ? "16-byte records over PCIe, re-expanded to SLAM::point (24 B) on the host inside the timed loop" : "packed 16-byte vslam_point16 records (vslam_points16_expand not timed)")),
                        hwq ? hwq : "default", placement.json().c_str());
        }
        vslam::BatchDetector::free_pinned(h_frames);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Stream: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
