// flight_fuzz.cpp -- sanitizer harness for the Arrow IPC reader / writer of flight.hip (host code only).
//
// Built on the CPU box by tests/test_flight_sanitize.py:
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -x c++ flight.hip flight_fuzz.cpp
// flight.hip is compiled as plain C++ against the stub index below, so every byte the parser hands to the "library"
// is touched under AddressSanitizer: an offset that escaped the bounds checks shows up as a report, not as luck.
// The registry HAS the dataset the seeds ask for, so mutated k / list offsets / buffer lengths reach the code past
// NotFound (reference shapes: internal/store/vector_search_exchange.go:63-124, store_lifecycle.go:66-76).
//
// usage: flight_fuzz <n_mutations> <rng_seed> <dim> request:<file> ... ingest:<file> ... must_fail_request:<file> ...
#include "../../include/longbow_gpu.h"

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

// ---- stub of the index calls flight.hip makes ---------------------------------------------------------------------
struct lb_gpu_index {
    int dim = 8;
    int64_t n = 0;
    uint64_t sink = 0; // checksum of everything read, so the reads cannot be optimised away
    int64_t searches = 0, adds = 0;
};
extern "C" {
int lb_gpu_index_dim(const lb_gpu_index *h) { return h ? h->dim : 0; }
int64_t lb_gpu_index_ntotal(const lb_gpu_index *h) { return h ? h->n : 0; }
const char *lb_gpu_status_string(int) { return "stub"; }
const char *lb_gpu_last_error(const lb_gpu_index *) { return ""; }
int lb_gpu_index_search(lb_gpu_index *h, int64_t nq, const float *queries, int k, float *dist, int64_t *labels)
{
    if (!h || nq < 0 || k <= 0 || !queries || !dist || !labels) return LB_ERR_INVALID_ARG;
    if (k > LB_MAX_K) return LB_ERR_UNSUPPORTED;
    const unsigned char *q = reinterpret_cast<const unsigned char *>(queries);
    for (int64_t i = 0; i < nq * h->dim * 4; i++) h->sink += q[i];
    for (int64_t i = 0; i < nq * k; i++) {
        dist[i] = (float)i;
        labels[i] = i < 3 ? i : -1; // three hits, the rest padding
    }
    h->searches++;
    return LB_OK;
}
int lb_gpu_index_add(lb_gpu_index *h, int64_t n, const float *vectors, const int64_t *ids)
{
    if (!h || n < 0 || (n > 0 && !vectors)) return LB_ERR_INVALID_ARG;
    const unsigned char *v = reinterpret_cast<const unsigned char *>(vectors);
    for (int64_t i = 0; i < n * h->dim * 4; i++) h->sink += v[i];
    if (ids)
        for (int64_t i = 0; i < n; i++) h->sink += (uint64_t)ids[i];
    h->n += n;
    h->adds++;
    return LB_OK;
}
}

static std::vector<uint8_t> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct Rng {
    uint64_t s;
    uint64_t next()
    {
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    size_t below(size_t n) { return n ? (size_t)(next() % n) : 0; }
};

static std::map<int, long> g_status;

static int run_request(lb_flight_datasets *reg, const std::vector<uint8_t> &in)
{
    // an exact-size heap copy: reads past the end are ASan reports
    uint8_t *copy = static_cast<uint8_t *>(std::malloc(in.size() ? in.size() : 1));
    std::memcpy(copy, in.data(), in.size());
    uint8_t *out = nullptr;
    size_t out_len = 0;
    char errbuf[256];
    const int rc = lb_flight_vector_search_exchange(reg, copy, in.size(), &out, &out_len, errbuf, sizeof errbuf);
    if (rc == 0) {
        uint64_t sum = 0;
        for (size_t i = 0; i < out_len; i++) sum += out[i]; // the response is readable over its whole length
        if (sum == 0xdeadbeefcafeull) std::puts("");
        lb_flight_free_buffer(out);
    }
    std::free(copy);
    g_status[rc]++;
    return rc;
}

static int run_ingest(lb_gpu_index *h, const std::vector<uint8_t> &in)
{
    uint8_t *copy = static_cast<uint8_t *>(std::malloc(in.size() ? in.size() : 1));
    std::memcpy(copy, in.data(), in.size());
    int64_t added = 0;
    char errbuf[256];
    const int rc = lb_flight_index_add_ipc(h, copy, in.size(), &added, errbuf, sizeof errbuf);
    std::free(copy);
    g_status[1000 + rc]++;
    return rc;
}

static void mutate(std::vector<uint8_t> &b, Rng &r)
{
    if (b.empty()) return;
    static const int64_t special[] = {0, 1, -1, 2, 7, 8, 9, 16, 63, 64, 255, 256, 2047, 2048, 2049, 4096, 65535, 65536,
                                      0x7fffffffll, 0x80000000ll, 0xffffffffll, (int64_t)1 << 40, (int64_t)1 << 61,
                                      ((int64_t)1 << 61) + 8, ((int64_t)1 << 62) + 1, INT64_MAX, INT64_MIN, -8, -4096};
    const int nspecial = (int)(sizeof special / sizeof special[0]);
    const int kind = (int)r.below(8);
    switch (kind) {
    case 0: case 1: { // flip a few bytes
        const int n = 1 + (int)r.below(4);
        for (int i = 0; i < n; i++) b[r.below(b.size())] ^= (uint8_t)(1u << r.below(8));
        break;
    }
    case 2: { // overwrite an aligned int32 with a special value
        if (b.size() < 4) break;
        const size_t at = r.below(b.size() / 4) * 4;
        const int32_t v = (int32_t)special[r.below(nspecial)];
        std::memcpy(&b[at], &v, 4);
        break;
    }
    case 3: case 4: { // overwrite an aligned int64 with a special value (offsets, lengths, row counts)
        if (b.size() < 8) break;
        const size_t at = r.below(b.size() / 8) * 8;
        const int64_t v = special[r.below(nspecial)];
        std::memcpy(&b[at], &v, 8);
        break;
    }
    case 5: // truncate
        b.resize(r.below(b.size()));
        break;
    case 6: { // random bytes over a short run
        const size_t at = r.below(b.size()), n = 1 + r.below(16);
        for (size_t i = at; i < b.size() && i < at + n; i++) b[i] = (uint8_t)r.next();
        break;
    }
    default: { // copy one aligned 16-byte block over another (swaps Buffer / FieldNode entries, vtables, offsets)
        if (b.size() < 32) break;
        const size_t a = r.below(b.size() / 16 - 1) * 16, c = r.below(b.size() / 16 - 1) * 16;
        std::memmove(&b[a], &b[c], 16);
        break;
    }
    }
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: flight_fuzz <n_mutations> <rng_seed> <dim> request:<file>|ingest:<file>|must_fail_request:<file> ...\n");
        return 2;
    }
    const long n_mut = std::atol(argv[1]);
    Rng rng{(uint64_t)std::atoll(argv[2])};
    lb_gpu_index idx;
    idx.dim = std::atoi(argv[3]);
    lb_flight_datasets *reg = lb_flight_datasets_new();
    lb_flight_datasets_put(reg, "ds", &idx); // the dataset the seeds ask for EXISTS: mutations get past NotFound
    std::vector<std::vector<uint8_t>> req, ing;
    for (int i = 4; i < argc; i++) {
        const std::string a = argv[i];
        if (a.rfind("request:", 0) == 0) {
            req.push_back(slurp(a.c_str() + 8));
            if (run_request(reg, req.back()) != 0) { std::fprintf(stderr, "seed %s is not accepted\n", a.c_str()); return 3; }
        } else if (a.rfind("ingest:", 0) == 0) {
            ing.push_back(slurp(a.c_str() + 7));
            if (run_ingest(&idx, ing.back()) != 0) { std::fprintf(stderr, "seed %s is not accepted\n", a.c_str()); return 3; }
        } else if (a.rfind("must_fail_request:", 0) == 0) {
            const auto bad = slurp(a.c_str() + 18);
            if (run_request(reg, bad) == 0) { std::fprintf(stderr, "%s was accepted\n", a.c_str()); return 4; }
            req.push_back(bad); // and a seed for further mutation
        } else if (a.rfind("must_fail_ingest:", 0) == 0) {
            const auto bad = slurp(a.c_str() + 17);
            if (run_ingest(&idx, bad) == 0) { std::fprintf(stderr, "%s was accepted\n", a.c_str()); return 4; }
            ing.push_back(bad);
        }
    }
    long done = 0;
    for (long i = 0; i < n_mut; i++) {
        const bool do_req = ing.empty() || (!req.empty() && (i & 1) == 0);
        const auto &seed = do_req ? req[rng.below(req.size())] : ing[rng.below(ing.size())];
        std::vector<uint8_t> m = seed;
        const int rounds = 1 + (int)rng.below(3);
        for (int r = 0; r < rounds; r++) mutate(m, rng);
        if (do_req) run_request(reg, m);
        else run_ingest(&idx, m);
        done++;
    }
    std::printf("mutations: %ld  searches reached: %" PRId64 "  adds reached: %" PRId64 "  sink %llu\n", done, idx.searches, idx.adds,
                (unsigned long long)idx.sink);
    for (auto &kv : g_status) std::printf("  status %d: %ld\n", kv.first, kv.second);
    lb_flight_datasets_free(reg);
    return 0;
}
