// Headless, asserting test of the C++ throughput host (batch_detector.hpp) - registered with CTest like the
// reference's target names.  Checks, all on the GPU through the C ABI:
//   1. the host-fed pipeline (submit / collect, several batches in flight, packed lists) returns, frame by
//      frame, exactly the lists the device-resident call leaves in HBM;
//   2. those lists equal what the per-image API of vslam_cxx.hpp produces for the same frame
//      (HarrisKeypoints; GaussPyramid + scaleSpaceCandidates per octave) - the drop-in functions and the
//      batched path are the same detector;
//   2b. with Options::describe the batch runs the rest of the DoG executable (FeaturePointLocalization, filterKeypoints,
//      SIFT) and returns per frame exactly what the per-image functions return, descriptors bit for bit;
//   3. the C ABI refuses an undersized output buffer with VSLAM_ERR_INVALID before launching anything
//      (include/vslam.h: vslam_batch_out carries buffer sizes), and a struct without struct_size.
//   usage: BatchDetector_Test [WxH] [frames]
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include "batch_detector.hpp"
#include "imgio.hpp"
#include "vslam_cxx.hpp"

static int failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) {                                                        \
            ++failures;                                                       \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                     \
    } while (0)

int main(int argc, char** argv) {
    try {
        int cols = 480, rows = 270, n = 6;
        if (argc > 1) std::sscanf(argv[1], "%dx%d", &cols, &rows);
        if (argc > 2) n = std::atoi(argv[2]);
        const size_t N = (size_t)rows * cols;
        uint8_t* h_frames = (uint8_t*)vslam::BatchDetector::alloc_pinned((size_t)n * N);
        std::vector<cv::Mat> imgs;
        for (int f = 0; f < n; ++f) {
            imgs.push_back(imgio::synthetic(rows, cols, f, 5));
            std::memcpy(h_frames + (size_t)f * N, imgs.back().data, N);
        }
        vslam::BatchDetector::Options opt;
        opt.rows = rows, opt.cols = cols, opt.batch = n, opt.slots = 3, opt.pipelines = 2;
        vslam::BatchDetector det(opt);
        const vslam_params& p = det.params();

        // ---- 1. host-fed: five batches through three slots, every result identical
        std::vector<vslam_kp> kp0;
        std::vector<vslam_point> pt0;
        std::vector<uint64_t> hoff, doff;
        int submitted = 0, collected = 0;
        const int total = 5;
        for (; submitted < 3; ++submitted) det.submit(h_frames, n);
        while (collected < total) {
            const vslam::BatchResult& r = det.collect();
            EXPECT(r.n_frames == n && !r.truncated);
            if (collected == 0) {
                kp0.assign(r.harris, r.harris + r.harris_records);
                pt0.assign(r.dog, r.dog + r.dog_records);
                hoff.assign(r.harris_offsets, r.harris_offsets + n + 1);
                doff.assign(r.dog_offsets, r.dog_offsets + n + 1);
                EXPECT(r.harris_records == hoff[n] && r.dog_records == doff[n]);
            } else {
                EXPECT(r.harris_records == kp0.size() && r.dog_records == pt0.size());
                EXPECT(std::memcmp(r.harris, kp0.data(), kp0.size() * sizeof(vslam_kp)) == 0);
                EXPECT(std::memcmp(r.dog, pt0.data(), pt0.size() * sizeof(vslam_point)) == 0);
                EXPECT(std::memcmp(r.harris_offsets, hoff.data(), (n + 1) * 8) == 0 && std::memcmp(r.dog_offsets, doff.data(), (n + 1) * 8) == 0);
            }
            ++collected;
            if (submitted < total) det.submit(h_frames, n), ++submitted;
        }
        EXPECT(det.in_flight() == 0);
        bool threw = false;
        try {
            det.collect();
        } catch (const vslam::Error&) {
            threw = true;
        }
        EXPECT(threw);

        // ---- device-resident call on the same frames: lists in HBM equal the host-fed ones
        uint8_t* d_frames = nullptr;
        EXPECT(hipMalloc((void**)&d_frames, (size_t)n * N) == hipSuccess);
        EXPECT(hipMemcpy(d_frames, h_frames, (size_t)n * N, hipMemcpyHostToDevice) == hipSuccess);
        det.detect_device(d_frames, N, n);
        det.sync();
        const vslam_batch_out& o = det.device_outputs();
        std::vector<uint32_t> hc(n), dc(n);
        EXPECT(hipMemcpy(hc.data(), o.harris_counts, 4 * n, hipMemcpyDeviceToHost) == hipSuccess);
        EXPECT(hipMemcpy(dc.data(), o.dog_counts, 4 * n, hipMemcpyDeviceToHost) == hipSuccess);
        uint64_t tot[2] = {0, 0}, sum[2] = {0, 0};
        EXPECT(hipMemcpy(tot, det.device_totals(), 16, hipMemcpyDeviceToHost) == hipSuccess);
        for (int f = 0; f < n; ++f) {
            sum[0] += hc[f], sum[1] += dc[f];
            EXPECT(hc[f] == hoff[f + 1] - hoff[f] && dc[f] == doff[f + 1] - doff[f]);
            std::vector<vslam_kp> k(hc[f]);
            std::vector<vslam_point> q(dc[f]);
            if (hc[f]) EXPECT(hipMemcpy(k.data(), o.harris_kps + (size_t)f * p.harris_cap, hc[f] * sizeof(vslam_kp), hipMemcpyDeviceToHost) == hipSuccess);
            if (dc[f]) EXPECT(hipMemcpy(q.data(), o.dog_points + (size_t)f * p.dog_cap, dc[f] * sizeof(vslam_point), hipMemcpyDeviceToHost) == hipSuccess);
            EXPECT(std::memcmp(k.data(), kp0.data() + hoff[f], k.size() * sizeof(vslam_kp)) == 0);
            EXPECT(std::memcmp(q.data(), pt0.data() + doff[f], q.size() * sizeof(vslam_point)) == 0);
        }
        EXPECT(tot[0] == sum[0] && tot[1] == sum[1] && sum[0] > 0 && sum[1] > 0);

        // ---- 1b. two pipelines: the kernels of consecutive batches overlap (own context, stream and image buffers each).
        //          Two different batches A, B alternate through three slots; every result equals what a one-pipeline
        //          detector gives for the same batch, host-fed and device-resident
        {
            EXPECT(det.pipelines() == 2);
            uint8_t* hB = (uint8_t*)vslam::BatchDetector::alloc_pinned((size_t)n * N);
            for (int f = 0; f < n; ++f) {
                cv::Mat b = imgio::synthetic(rows, cols, f, 11);
                std::memcpy(hB + (size_t)f * N, b.data, N);
            }
            vslam::BatchDetector::Options o1 = opt;
            o1.pipelines = 1;
            vslam::BatchDetector one(o1);
            EXPECT(one.pipelines() == 1);
            one.submit(hB, n);
            const vslam::BatchResult& rb = one.collect();
            const std::vector<vslam_kp> kpB(rb.harris, rb.harris + rb.harris_records);
            const std::vector<vslam_point> ptB(rb.dog, rb.dog + rb.dog_records);
            EXPECT(kpB.size() != kp0.size() || std::memcmp(kpB.data(), kp0.data(), kpB.size() * sizeof(vslam_kp)) != 0);  // B is another batch
            const uint8_t* src[2] = {h_frames, hB};
            int sub = 0, col = 0;
            const int tot2 = 7;
            for (; sub < 3; ++sub) det.submit(src[sub & 1], n);
            while (col < tot2) {
                const vslam::BatchResult& r = det.collect();
                const std::vector<vslam_kp>& wk = (col & 1) ? kpB : kp0;
                const std::vector<vslam_point>& wp = (col & 1) ? ptB : pt0;
                EXPECT(r.harris_records == wk.size() && r.dog_records == wp.size());
                EXPECT(r.harris_records == wk.size() && std::memcmp(r.harris, wk.data(), wk.size() * sizeof(vslam_kp)) == 0);
                EXPECT(r.dog_records == wp.size() && std::memcmp(r.dog, wp.data(), wp.size() * sizeof(vslam_point)) == 0);
                ++col;
                if (sub < tot2) det.submit(src[sub & 1], n), ++sub;
            }
            // a result outlives the next submit(): with slots - 1 batches in flight the slot just collected is the one the
            // next submit() takes, and its pack-stream copies used to land in the offsets / counts the caller still held
            {
                det.submit(h_frames, n);                       // A
                det.submit(hB, n);                             // B
                const vslam::BatchResult& ra = det.collect();  // A: its slot is free again
                det.submit(hB, n);                             // B into A's slot
                det.sync();                                    // ... and all of it has run
                EXPECT(ra.n_frames == n && ra.harris_records == kp0.size() && ra.dog_records == pt0.size());
                for (int f = 0; f <= n; ++f) EXPECT(ra.harris_offsets[f] == hoff[f] && ra.dog_offsets[f] == doff[f]);
                for (int f = 0; f < n; ++f) EXPECT(ra.harris_counts[f] == hoff[f + 1] - hoff[f] && ra.dog_counts[f] == doff[f + 1] - doff[f]);
                const vslam::FrameKeypoints fk = ra.frame(n - 1);
                EXPECT(fk.n_harris == hoff[n] - hoff[n - 1] && fk.n_dog == doff[n] - doff[n - 1] && fk.harris_total == fk.n_harris && fk.dog_total == fk.n_dog);
                det.collect();
                det.collect();
                EXPECT(det.in_flight() == 0);
            }
            // device-resident: A and B back to back without a sync in between, each on its own pipeline
            uint8_t* dAB = nullptr;
            EXPECT(hipMalloc((void**)&dAB, 2 * (size_t)n * N) == hipSuccess);
            EXPECT(hipMemcpy(dAB, h_frames, (size_t)n * N, hipMemcpyHostToDevice) == hipSuccess);
            EXPECT(hipMemcpy(dAB + (size_t)n * N, hB, (size_t)n * N, hipMemcpyHostToDevice) == hipSuccess);
            const uint64_t* tp[4];
            const void* st[4];
            const vslam_point* lists[4];
            for (int k = 0; k < 4; ++k) {
                EXPECT(det.next_stream() != det.stream() || k == 0);
                det.detect_device(dAB + (size_t)(k & 1) * n * N, N, n);
                tp[k] = det.device_totals(), st[k] = det.stream(), lists[k] = det.device_outputs().dog_points;
            }
            det.sync();
            EXPECT(st[0] != st[1] && st[0] == st[2] && st[1] == st[3] && tp[0] != tp[1] && lists[0] != lists[1] && lists[0] == lists[2]);
            uint64_t ta[2], tb[2];
            EXPECT(hipMemcpy(ta, tp[2], 16, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(tb, tp[3], 16, hipMemcpyDeviceToHost) == hipSuccess);
            uint64_t wa[2] = {0, 0}, wb[2] = {0, 0};
            for (int f = 0; f < n; ++f) wa[0] += hoff[f + 1] - hoff[f], wa[1] += doff[f + 1] - doff[f], wb[0] += rb.harris_counts[f], wb[1] += rb.dog_counts[f];
            EXPECT(ta[0] == wa[0] && ta[1] == wa[1] && tb[0] == wb[0] && tb[1] == wb[1]);
            std::vector<vslam_point> q(rb.dog_counts[n - 1]);  // the last frame's list of batch B in HBM
            if (!q.empty()) {
                EXPECT(hipMemcpy(q.data(), lists[3] + (size_t)(n - 1) * p.dog_cap, q.size() * sizeof(vslam_point), hipMemcpyDeviceToHost) == hipSuccess);
                EXPECT(std::memcmp(q.data(), rb.dog + rb.dog_offsets[n - 1], q.size() * sizeof(vslam_point)) == 0);
            }
            (void)hipFree(dAB);
            vslam::BatchDetector::free_pinned(hB);
        }

        // ---- 1c. Options::compact_points: the DoG list crosses PCIe as 16-byte records (vslam_pack_points16_dev) and
        //          BatchResult::frame() rebuilds the SLAM::points - byte for byte the 24-byte records of the plain detector
        {
            vslam::BatchDetector::Options oc = opt;
            oc.pipelines = 1, oc.compact_points = true;
            vslam::BatchDetector cdet(oc);
            cdet.submit(h_frames, n);
            const vslam::BatchResult& rc = cdet.collect();
            EXPECT(rc.dog == nullptr && rc.dog16 != nullptr && rc.dog_records == pt0.size() && rc.harris_records == kp0.size() && !rc.truncated);
            EXPECT(std::memcmp(rc.harris, kp0.data(), kp0.size() * sizeof(vslam_kp)) == 0);
            uint64_t seen = 0;
            for (int f = 0; f < n; ++f) {
                const vslam::FrameKeypoints fk = rc.frame(f);
                EXPECT(fk.n_dog == doff[f + 1] - doff[f] && fk.dog == fk.dog_expanded.data());
                EXPECT(fk.n_dog == 0 || std::memcmp(fk.dog, pt0.data() + doff[f], fk.n_dog * sizeof(vslam_point)) == 0);
                seen += fk.n_dog;
                // a copy owns its own expanded list (and outlives the original); a moved one keeps the buffer
                vslam::FrameKeypoints copy;
                {
                    vslam::FrameKeypoints tmp = rc.frame(f);
                    copy = tmp;
                    EXPECT(copy.dog == copy.dog_expanded.data() && (copy.n_dog == 0 || copy.dog != tmp.dog));
                }
                EXPECT(copy.n_dog == fk.n_dog && (copy.n_dog == 0 || std::memcmp(copy.dog, fk.dog, fk.n_dog * sizeof(vslam_point)) == 0));
                const vslam_point* before = copy.dog;
                const vslam::FrameKeypoints moved = std::move(copy);
                EXPECT(moved.dog == before && moved.dog == moved.dog_expanded.data());
            }
            EXPECT(seen == pt0.size() && seen > 0);
        }

        // ---- 2. the per-image drop-in functions give the same lists (frames 0 and n-1)
        for (int f : {0, n - 1}) {
            const std::vector<vslam_kp> kps = HarrisKeypoints(imgs[f], 0.04f);
            EXPECT(kps.size() == hoff[f + 1] - hoff[f]);
            EXPECT(kps.empty() || std::memcmp(kps.data(), kp0.data() + hoff[f], kps.size() * sizeof(vslam_kp)) == 0);
            GaussPyramid pyramid{imgs[f], p.n_octaves, p.sigma0};
            std::vector<SLAM::point> cand;
            for (int oc = 0; oc < pyramid.getNumOctaves(); ++oc) scaleSpaceCandidates(cand, pyramid, oc, p.extrema_window, p.min_contrast);
            EXPECT(cand.size() == doff[f + 1] - doff[f]);
            EXPECT(cand.empty() || std::memcmp(cand.data(), pt0.data() + doff[f], cand.size() * sizeof(vslam_point)) == 0);
        }

        // ---- 2b. the rest of the DoG executable in the batch (Options::describe: localize + filterKeypoints + SIFT):
        //          frame by frame the same oriented points, descriptors and defined flags as the per-image functions
        {
            const int m = 3;
            std::vector<cv::Mat> im2;
            uint8_t* h2 = (uint8_t*)vslam::BatchDetector::alloc_pinned((size_t)m * N);
            for (int f = 0; f < m; ++f) {
                cv::Mat a = imgio::synthetic(rows, cols, f, 9);
                if (f == 1)  // a uniform-noise frame: thousands of oriented points (the checkerboard has a few dozen)
                    for (int r = 0; r < rows; ++r)
                        for (int c = 0; c < cols; ++c) a.at<cv::uchar>(r, c) = (cv::uchar)(imgio::splitmix64(0x5EED0009ull ^ (uint64_t)((uint64_t)r * cols + c)) & 255);
                im2.push_back(a);
                std::memcpy(h2 + (size_t)f * N, a.data, N);
            }
            vslam::BatchDetector::Options o2 = opt;
            o2.batch = m, o2.describe = true, o2.slots = 2;
            o2.host_oriented_per_frame = (size_t)1 << 16, o2.host_descriptors_per_frame = (size_t)1 << 16;
            vslam::BatchDetector d2(o2);
            EXPECT(d2.params().localize == 1 && d2.params().orient == 1);
            d2.submit(h2, m);
            const vslam::BatchResult& r2 = d2.collect();
            EXPECT(r2.n_frames == m && !r2.truncated && r2.oriented_records > (uint64_t)(N / 200) && r2.descriptor_records == r2.oriented_records);
            for (int f = 0; f < m; ++f) {
                const vslam::FrameKeypoints k = r2.frame(f);
                GaussPyramid pyramid{im2[f], d2.params().n_octaves, d2.params().sigma0};
                std::vector<SLAM::point> kps_all, ori_all;
                std::vector<std::vector<float>> desc_all;
                std::vector<unsigned char> def_all;
                for (int oc = 0; oc < pyramid.getNumOctaves(); ++oc) {
                    std::vector<SLAM::point> kps, ori;
                    initialKeypointDetection(kps, pyramid, oc, d2.params().extrema_window);  // :785
                    filterKeypoints(pyramid, oc, kps, ori);                                  // :787
                    std::vector<unsigned char> def;
                    SIFT(ori, desc_all, pyramid, oc, &def);                                  // :791
                    kps_all.insert(kps_all.end(), kps.begin(), kps.end());
                    ori_all.insert(ori_all.end(), ori.begin(), ori.end());
                    def_all.insert(def_all.end(), def.begin(), def.end());
                }
                EXPECT(k.n_dog == kps_all.size() && (kps_all.empty() || std::memcmp(k.dog, kps_all.data(), kps_all.size() * sizeof(vslam_point)) == 0));
                EXPECT(k.n_oriented == ori_all.size() && k.oriented_total == ori_all.size());
                EXPECT(ori_all.empty() || std::memcmp(k.oriented, ori_all.data(), ori_all.size() * sizeof(vslam_point)) == 0);
                EXPECT(k.descriptors != nullptr && k.descriptor_defined != nullptr && desc_all.size() == ori_all.size());
                size_t bad = 0;
                for (size_t q = 0; q < ori_all.size() && k.descriptors; ++q) {
                    bad += (k.descriptor_defined[q] != 0) != (def_all[q] != 0);
                    bad += std::memcmp(k.descriptors + q * 128, desc_all[q].data(), 128 * sizeof(float)) != 0;  // bit patterns (NaN descriptors included)
                }
                EXPECT(bad == 0);
            }
            vslam::BatchDetector::free_pinned(h2);
        }

        // ---- 3. undersized buffers are refused by the C ABI itself
        vslam_batch_out bad = o;
        bad.pyramid_bytes -= 1;
        int rc = vslam_detect_batch_dev(det.context(), &p, d_frames, N, n, &bad);
        EXPECT(rc == VSLAM_ERR_INVALID && std::strstr(vslam_last_error(det.context()), "pyramid") != nullptr);
        bad = o;
        bad.dog_points_bytes = (size_t)n * p.dog_cap * sizeof(vslam_point) - sizeof(vslam_point);
        EXPECT(vslam_detect_batch_dev(det.context(), &p, d_frames, N, n, &bad) == VSLAM_ERR_INVALID);
        bad = o;
        bad.struct_size = 0;
        EXPECT(vslam_detect_batch_dev(det.context(), &p, d_frames, N, n, &bad) == VSLAM_ERR_INVALID);
        bad = o;  // a larger buffer than needed is fine
        bad.pyramid_bytes += 4096;
        EXPECT(vslam_detect_batch_dev(det.context(), &p, d_frames, N, n, &bad) == VSLAM_OK);
        det.sync();
        (void)hipFree(d_frames);
        vslam::BatchDetector::free_pinned(h_frames);
        std::printf("{\"exe\": \"BatchDetector_Test\", \"rows\": %d, \"cols\": %d, \"frames\": %d, \"harris\": %llu, \"dog\": %llu, \"failures\": %d}\n", rows, cols,
                    n, (unsigned long long)sum[0], (unsigned long long)sum[1], failures);
        return failures ? 2 : 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "BatchDetector_Test: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
