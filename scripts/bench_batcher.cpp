// bench_batcher.cpp -- concurrent single-query callers (the reference's calling pattern: one
// ValidatedVectorReadIndex::search per operator invocation, many tokio tasks) with and without the batching operator.
//
//   g++ -O2 -std=c++17 -I include scripts/bench_batcher.cpp -o scripts/_bin/bench_batcher \
//       -L helix-db_amd -lhelix_vec_gfx950 -Wl,-rpath,$PWD/helix-db_amd -lpthread
//   scripts/_bin/bench_batcher <dir with the arrays written by scripts/bench_batcher.py> <threads> <queries per thread>
#include <linux/futex.h>
#include <poll.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <climits>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "helix_vec.h"

template <typename T> static std::vector<T> load(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)bytes / sizeof(T));
    if (fread(v.data(), 1, (size_t)bytes, f) != (size_t)bytes) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <dir> <threads> <queries per thread> [strict|default] [lanes]\n", argv[0]); return 2; }
    hvx_runtime_prepare(0); // hardware queues for the dispatcher lanes: before the process's first HIP call (include/helix_vec.h)
    const std::string dir = argv[1];
    const int threads = atoi(argv[2]), per = atoi(argv[3]);
    const bool strict = argc > 4 && !strcmp(argv[4], "strict");
    const uint32_t lanes = argc > 5 ? (uint32_t)atoi(argv[5]) : 2u;
    const bool direct = !(argc > 6 && !strcmp(argv[6], "nodirect")); // the one-call-per-query baseline takes ~1 min at 1 024 threads
    auto meta = load<uint64_t>(dir + "/meta.u64"); // n, dim, m, entry, max_layer
    const uint64_t n = meta[0];
    const uint32_t dim = (uint32_t)meta[1], m = (uint32_t)meta[2];
    auto ids = load<uint64_t>(dir + "/ids.u64");
    auto vec = load<float>(dir + "/vectors.f32");
    auto l0o = load<uint64_t>(dir + "/l0_offsets.u64");
    auto l0n = load<uint64_t>(dir + "/l0_neighbors.u64");
    auto lvl = load<uint16_t>(dir + "/level.u16");
    auto upo = load<uint64_t>(dir + "/up_offsets.u64");
    auto upn = load<uint64_t>(dir + "/up_neighbors.u64");
    auto qs = load<float>(dir + "/queries.f32");
    const uint32_t nq = (uint32_t)(qs.size() / dim);
    hvx_index_desc d;
    memset(&d, 0, sizeof(d));
    d.dim = dim; d.metric = HVX_L2_SQUARED; d.dtype = HVX_F32; d.float_kernel = HVX_KERNEL_AVX_FMA;
    d.n = n; d.m = m; d.m0 = 2 * m; d.has_entry = 1; d.max_layer = (uint32_t)meta[4]; d.entry_point = meta[3];
    d.device = -1; d.max_batch = 1024;
    hvx_index *ix = nullptr;
    if (hvx_index_import(&d, ids.data(), vec.data(), l0o.data(), l0n.data(), lvl.data(), upo.data(), upn.data(), &ix)) {
        fprintf(stderr, "import failed: %s\n", hvx_last_error());
        return 1;
    }
    hvx_search_params p;
    hvx_search_params_default(&p, 10);
    if (strict) { p.simhash_mode = HVX_SIMHASH_OFF; p.pre_simhash_sampling_ratio_override = 1.0f; }
    else {
        hvx_simhash_config c;
        hvx_simhash_config_default(&c);
        if (hvx_index_set_simhash(ix, &c, nullptr)) { fprintf(stderr, "%s\n", hvx_last_error()); return 1; }
    }
    const uint32_t k = p.k;
    // BATCHER_RESTRICTED=<m>: the operator's prefiltered branch (storage.rs:140-163 `Some(candidates)`): every call brings ITS OWN m candidate
    // ids (a pool of 4 096 random lists; callers walk through it), batched by hvx_batcher_new_restricted; the direct baseline = one
    // hvx_search_restricted_batch_params call per query
    const char *e_res = getenv("BATCHER_RESTRICTED");
    const uint32_t res_m = e_res ? (uint32_t)atoi(e_res) : 0u;
    const uint32_t n_lists = 4096;
    std::vector<uint64_t> lists;
    hvx_restricted_params rp;
    hvx_restricted_params_default(&rp, k, 100);
    if (res_m) {
        lists.resize((size_t)n_lists * res_m);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (auto &v : lists) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = ids[x % n]; }
    }
    auto list_of = [&](long call) { return lists.data() + (size_t)((uint64_t)call % n_lists) * res_m; };
    auto run = [&](hvx_batcher *bt, double *qps, double *mean_us, double *p99_us) {
        std::vector<std::vector<double>> lat(threads);
        std::atomic<int> failures{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] {
                std::vector<uint64_t> oi(k);
                std::vector<float> os(k);
                uint32_t cnt = 0, st = 0;
                for (int i = 0; i < per; ++i) {
                    const float *q = qs.data() + (size_t)((t * per + i) % nq) * dim;
                    const auto a = std::chrono::steady_clock::now();
                    int rc;
                    if (res_m) rc = bt ? hvx_batcher_search_restricted(bt, q, list_of((long)t * per + i), res_m, oi.data(), os.data(), &cnt)
                                       : hvx_search_restricted_batch_params(ix, q, 1, &rp, list_of((long)t * per + i), nullptr, res_m, oi.data(), os.data(), &cnt, &st, nullptr, nullptr);
                    else rc = bt ? hvx_batcher_search(bt, q, oi.data(), os.data(), &cnt)
                                 : hvx_search_batch_params(ix, q, 1, &p, oi.data(), os.data(), &cnt, &st, nullptr, nullptr, nullptr);
                    if (rc || cnt != k) failures++;
                    lat[t].push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
                }
            });
        for (auto &x : th) x.join();
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<double> all;
        for (auto &v : lat) all.insert(all.end(), v.begin(), v.end());
        std::sort(all.begin(), all.end());
        double sum = 0;
        for (double x : all) sum += x;
        *qps = all.size() / secs;
        *mean_us = sum / all.size();
        *p99_us = all[(size_t)(all.size() * 0.99)];
        if (failures) fprintf(stderr, "%d failed calls\n", failures.load());
    };
    // the non-blocking form (hvx_batcher_submit / _poll + eventfd): `subs` submitter threads, each keeping up to `window` tickets in
    // flight and sleeping in poll(2) on the batcher's eventfd between completions -- what a tokio reactor does with AsyncFd
    auto run_nb = [&](hvx_batcher *bt, int subs, int window, long total, double *qps, double *mean_us, double *p99_us) {
        const int fd = hvx_batcher_eventfd(bt);
        std::vector<std::vector<double>> lat(subs);
        std::atomic<int> failures{0};
        std::atomic<long> busy{0};
        // the reactor: ONE thread owns the eventfd (as tokio's driver does) and turns every tick into a generation bump the submitters
        // sleep on -- a submitter that was busy polling its tickets when the tick came cannot miss it (it re-reads the generation)
        std::atomic<uint32_t> gen{0};
        std::atomic<bool> reactor_stop{false};
        auto fwait = [](std::atomic<uint32_t> *a, uint32_t expect, long us) {
            timespec ts{us / 1000000, (us % 1000000) * 1000};
            syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAIT_PRIVATE, expect, &ts, nullptr, 0);
        };
        std::thread reactor([&] {
            while (!reactor_stop.load()) {
                pollfd pf{fd, POLLIN, 0};
                if (poll(&pf, 1, 2) > 0) {
                    uint64_t v;
                    (void)!read(fd, &v, sizeof(v));
                    gen.fetch_add(1);
                    syscall(SYS_futex, reinterpret_cast<uint32_t *>(&gen), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
                }
            }
        });
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < subs; ++t)
            th.emplace_back([&, t] {
                const long quota = total / subs;
                std::vector<hvx_batcher_ticket> tk(window);
                std::vector<std::chrono::steady_clock::time_point> ts(window);
                std::vector<char> live(window, 0);
                std::vector<uint64_t> oi(k);
                std::vector<float> os(k);
                long issued = 0, done = 0;
                int inflight = 0;
                lat[t].reserve(quota);
                while (done < quota) {
                    const uint32_t g0 = gen.load();
                    bool progressed = false;
                    for (int w = 0; w < window; ++w) {
                        if (!live[w]) continue;
                        uint32_t cnt = 0;
                        const int rc = hvx_batcher_poll(bt, &tk[w], oi.data(), os.data(), &cnt);
                        if (rc == HVX_PENDING) continue;
                        if (rc || cnt != k) failures++;
                        lat[t].push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ts[w]).count());
                        live[w] = 0; --inflight; ++done;
                        progressed = true;
                    }
                    for (int w = 0; w < window && issued < quota && inflight < window; ++w) {
                        if (live[w]) continue;
                        const float *q = qs.data() + (size_t)((t * quota + issued) % nq) * dim;
                        ts[w] = std::chrono::steady_clock::now();
                        const int rc = res_m ? hvx_batcher_submit_restricted(bt, q, list_of((long)t * quota + issued), res_m, &tk[w]) : hvx_batcher_submit(bt, q, &tk[w]);
                        if (rc == HVX_ERR_BUSY) { busy++; break; }
                        if (rc) { failures++; break; }
                        live[w] = 1; ++inflight; ++issued;
                        progressed = true;
                    }
                    if (!progressed) fwait(&gen, g0, 500); // nothing completed, nothing could be submitted: until the next tick
                }
            });
        for (auto &x : th) x.join();
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        reactor_stop.store(true);
        reactor.join();
        std::vector<double> all;
        for (auto &v : lat) all.insert(all.end(), v.begin(), v.end());
        std::sort(all.begin(), all.end());
        double sum = 0;
        for (double x : all) sum += x;
        *qps = all.size() / secs;
        *mean_us = sum / all.size();
        *p99_us = all[(size_t)(all.size() * 0.99)];
        if (failures) fprintf(stderr, "%d failed calls\n", failures.load());
        fprintf(stderr, "non-blocking: %d submitters x %d tickets + 1 reactor thread on the eventfd, %ld submits answered BUSY\n", subs, window, busy.load());
    };
    auto cgroup = [](unsigned long long out[3]) { // nr_throttled, throttled_usec, usage_usec of this container (cgroup v2)
        out[0] = out[1] = out[2] = 0;
        FILE *f = fopen("/sys/fs/cgroup/cpu.stat", "r");
        if (!f) return;
        char key[64]; unsigned long long v;
        while (fscanf(f, "%63s %llu", key, &v) == 2) {
            if (!strcmp(key, "nr_throttled")) out[0] = v;
            else if (!strcmp(key, "throttled_usec")) out[1] = v;
            else if (!strcmp(key, "usage_usec")) out[2] = v;
        }
        fclose(f);
    };
    double q0, m0, p0, q1, m1, p1;
    q0 = m0 = p0 = 0;
    if (direct) run(nullptr, &q0, &m0, &p0); // direct one-query calls on one handle: every call is a launch + two PCIe copies
    hvx_batcher *bt = nullptr;
    // harness knobs (the library reads no environment; lanes inherit the parent handle's settings): BATCHER_OCC = queries per SIMD of
    // the lanes' kernels, BATCHER_PAIR = HVX_OPT_HNSW_PAIR, BATCHER_WAIT = max_wait_us
    const char *e_occ = getenv("BATCHER_OCC"), *e_pair = getenv("BATCHER_PAIR"), *e_wait = getenv("BATCHER_WAIT");
    const uint32_t wait_us = e_wait ? (uint32_t)atoi(e_wait) : 100u;
    if (e_occ && hvx_index_set_occupancy(ix, (uint32_t)atoi(e_occ))) { fprintf(stderr, "%s\n", hvx_last_error()); return 1; }
    if (e_pair && hvx_index_set_option(ix, HVX_OPT_HNSW_PAIR, (uint32_t)atoi(e_pair))) { fprintf(stderr, "%s\n", hvx_last_error()); return 1; }
    const char *e_maxb = getenv("BATCHER_MAXB"); // (default 1 024 = the index's max_batch; small values exercise full batches and void claims)
    if (res_m ? hvx_batcher_new_restricted(ix, &rp, e_maxb ? (uint32_t)atoi(e_maxb) : 1024u, wait_us, lanes, res_m, &bt)
              : hvx_batcher_new_lanes(ix, &p, e_maxb ? (uint32_t)atoi(e_maxb) : 1024u, wait_us, lanes, &bt)) { fprintf(stderr, "%s\n", hvx_last_error()); return 1; }
    const char *e_nb = getenv("BATCHER_NB"); // "<submitters>x<tickets>": the non-blocking form instead of one blocked thread per query
    int nb_subs = 0, nb_win = 0;
    if (e_nb && sscanf(e_nb, "%dx%d", &nb_subs, &nb_win) != 2) { fprintf(stderr, "BATCHER_NB=<submitters>x<tickets>\n"); return 2; }
    const long nb_total = (long)threads * per;
    if (nb_subs) run_nb(bt, nb_subs, nb_win, nb_total, &q1, &m1, &p1); else run(bt, &q1, &m1, &p1);
    unsigned long long cg0[3], cg1[3];
    cgroup(cg0);
    const auto tb0 = std::chrono::steady_clock::now();
    if (nb_subs) run_nb(bt, nb_subs, nb_win, nb_total, &q1, &m1, &p1); else run(bt, &q1, &m1, &p1);
    const double run_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count();
    cgroup(cg1);
    fprintf(stderr, "timed run: %.3f s wall, %.2f CPU cores busy on average, cgroup throttled %llu time(s) for %.1f ms\n", run_s,
            (cg1[2] - cg0[2]) / 1e6 / run_s, cg1[0] - cg0[0], (cg1[1] - cg0[1]) / 1e3);
    uint64_t nb = 0, nqs = 0, nf = 0;
    hvx_batcher_stats(bt, &nb, &nqs, &nf);
    hvx_batcher_times tm;
    hvx_batcher_lane_times(bt, &tm);
    const double tot = (double)(tm.idle_ns + tm.collect_ns + tm.drain_ns + tm.fill_ns + tm.device_ns + tm.wake_ns) + 1.0;
    fprintf(stderr, "lane time shares (%u lanes): idle %.3f collect %.3f drain %.3f fill %.3f device %.3f wake %.3f; device ms per batch %.3f\n", tm.lanes,
            tm.idle_ns / tot, tm.collect_ns / tot, tm.drain_ns / tot, tm.fill_ns / tot, tm.device_ns / tot, tm.wake_ns / tot, nb ? tm.device_ns / 1e6 / nb : 0.0);
    printf("{\"workload\": \"%llu x %u f32, %s, k=10, %d caller threads x %d single-query calls\", "
           "\"direct_calls\": {\"qps\": %.0f, \"mean_us\": %.1f, \"p99_us\": %.1f}, "
           "\"batcher\": {\"lanes\": %u, \"qps\": %.0f, \"mean_us\": %.1f, \"p99_us\": %.1f, \"mean_batch\": %.1f, \"max_wait_us\": %u, \"occ\": \"%s\", \"pair\": \"%s\"}}\n",
           (unsigned long long)n, dim, res_m ? (std::string("restricted, every call its own ") + std::to_string(res_m) + " candidate ids (exact)").c_str() : (strict ? "strict ef=100" : "SearchParams::new(10)"), threads, per, q0, m0, p0, lanes, q1, m1, p1,
           nb ? (double)nqs / nb : 0.0, wait_us, e_occ ? e_occ : "default", e_pair ? e_pair : "default");
    hvx_batcher_free(bt);
    hvx_index_free(ix);
    return 0;
}
