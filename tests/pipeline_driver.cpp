// The batched-video mode from C++, without Python or PyTorch: what an application that links liborbfe.so writes (include/orbfe.h,
// "the batched-video mode").  TEST INFRASTRUCTURE (tests/test_pipeline_gpu.py compiles and runs it and compares what it writes with
// the Python wrapper's results, byte for byte):
//
//     pipeline_driver <frames.u8 | directory of .pgm> <frames> <rows> <cols> <steps> <out.records> [force_gather]
//         one rank: `steps` steps over the batch (a looping stream), the last record set (or, force_gather = 1, the block a communicator
//         of one rank gathered) + the match counts to <out.records>
//
//     pipeline_driver <frames.u8> <frames> <rows> <cols> <steps> <out prefix> gather <rank> <world> <id file> [every_step]
//         rank `rank` of `world` processes (ORBFE_RCCL_LIB = tests/fake_rccl.cpp puts them on one GPU): the file holds `nb` batches
//         of `frames` frames, step s runs batch s % nb, so every step's records differ.  Rank 0 is the destination: after enqueueing
//         step s it waits for the gather of step s - 1 (orbfe_pipeline_gathered_wait) -- step s is in flight meanwhile -- and writes
//         every rank's block of that batch to <out prefix>.step<s-1>.rank<r>, then releases the set.
//         every_step = 1 (no communicator needed: world 1, rank 0): synchronise after every step and write the rank's OWN record set
//         to <out prefix>.own.step<s> -- what the gathered blocks of a gather run must equal.
//
// A directory instead of a file: its *.pgm files (binary P5, 8 bit, all rows x cols) in name order are the stream -- the reference is
// validated on videos (Examples/Monocular/mono_cvcam.cc:128-148 reads frames one by one); this is that loop in batches.
#include <dirent.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/orbfe.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s: %d (%s)\n", #call, rc_, orbfe_last_error()); return 1; } } while (0)

static bool read_pgm(const std::string& path, int rows, int cols, uint8_t* dst)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int w = 0, h = 0, maxv = 0;
    auto token = [&](int* v) { // header tokens, '#' comments allowed
        int c;
        for (;;) {
            c = fgetc(f);
            if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
            if (c != ' ' && c != '\t' && c != '\n' && c != '\r') break;
        }
        ungetc(c, f);
        return fscanf(f, "%d", v) == 1;
    };
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '5' && token(&w) && token(&h) && token(&maxv) && w == cols && h == rows && maxv == 255;
    if (ok) { fgetc(f); ok = fread(dst, 1, (size_t)rows * cols, f) == (size_t)rows * cols; }
    fclose(f);
    return ok;
}

// every frame of the stream, tightly packed; returns the number of frames (0: error)
static size_t load_stream(const char* path, int rows, int cols, std::vector<uint8_t>& out)
{
    struct stat st;
    if (stat(path, &st) != 0) return 0;
    const size_t fb = (size_t)rows * cols;
    if (S_ISDIR(st.st_mode)) {
        std::vector<std::string> names;
        if (DIR* d = opendir(path)) {
            while (dirent* e = readdir(d)) {
                const std::string n = e->d_name;
                if (n.size() > 4 && n.substr(n.size() - 4) == ".pgm") names.push_back(std::string(path) + "/" + n);
            }
            closedir(d);
        }
        std::sort(names.begin(), names.end());
        out.resize(names.size() * fb);
        for (size_t i = 0; i < names.size(); i++)
            if (!read_pgm(names[i], rows, cols, out.data() + i * fb)) { fprintf(stderr, "%s: not a %d x %d 8-bit P5 file\n", names[i].c_str(), cols, rows); return 0; }
        return names.size();
    }
    const size_t n = (size_t)st.st_size / fb;
    out.resize(n * fb);
    FILE* f = fopen(path, "rb");
    if (!f || fread(out.data(), 1, out.size(), f) != out.size()) return 0;
    fclose(f);
    return n;
}

static int write_file(const std::string& path, const void* a, size_t na, const void* b = nullptr, size_t nb = 0)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return 5;
    fwrite(a, 1, na, f);
    if (b) fwrite(b, 1, nb, f);
    fclose(f);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 7) return 2;
    const int B = atoi(argv[2]), rows = atoi(argv[3]), cols = atoi(argv[4]), steps = atoi(argv[5]);
    const bool gather_mode = argc > 10 && !strcmp(argv[7], "gather");
    const int rank = gather_mode ? atoi(argv[8]) : 0, world = gather_mode ? atoi(argv[9]) : 1;
    const bool every_step = gather_mode && argc > 11 && atoi(argv[11]);
    const size_t pitch = ((size_t)cols + 63) / 64 * 64, fb = (size_t)rows * cols;
    std::vector<uint8_t> host;
    const size_t nframes = load_stream(argv[1], rows, cols, host);
    if (nframes < (size_t)B) { fprintf(stderr, "%s: fewer than %d frames of %d x %d\n", argv[1], B, cols, rows); return 3; }
    const int nb = (int)(nframes / (size_t)B);   // whole batches in the stream
    std::vector<uint8_t> padded((size_t)nb * B * rows * pitch, 0);
    for (size_t r = 0; r < (size_t)nb * B * rows; r++) std::copy(host.begin() + r * cols, host.begin() + (r + 1) * cols, padded.begin() + r * pitch);
    (void)fb;
    uint8_t* d_frames = nullptr;
    CHECK((int)hipMalloc(&d_frames, padded.size()));
    CHECK((int)hipMemcpy(d_frames, padded.data(), padded.size(), hipMemcpyHostToDevice));
    const size_t batch_bytes = (size_t)B * rows * pitch;

    orbfe_pipeline_config cfg;
    CHECK(orbfe_pipeline_config_default(&cfg, B, rows, cols));
    orbfe_pipeline* p = orbfe_pipeline_create(&cfg);
    if (!p) { fprintf(stderr, "orbfe_pipeline_create: %s\n", orbfe_last_error()); return 1; }
    orbfe_record_layout lay;
    CHECK(orbfe_pipeline_layout(p, &lay));
    std::vector<uint8_t> rec(lay.nbytes);

    if (gather_mode) {
        const std::string prefix = argv[6];
        if (!every_step) {
            // the communicator: rank 0 makes the id and leaves it in the id file, the others wait for the file
            uint8_t id[128];
            const std::string idf = argv[10], tmp = idf + ".tmp";
            if (rank == 0) {
                CHECK(orbfe_pipeline_comm_unique_id(id));
                if (write_file(tmp, id, 128) || rename(tmp.c_str(), idf.c_str()) != 0) return 5;
            } else {
                FILE* f = nullptr;
                for (int t = 0; t < 6000 && !(f = fopen(idf.c_str(), "rb")); t++) usleep(10000);
                if (!f || fread(id, 1, 128, f) != 128) { fprintf(stderr, "no id file\n"); return 6; }
                fclose(f);
            }
            CHECK(orbfe_pipeline_comm_init(p, id, rank, world, 0));
        }
        std::vector<int32_t> sets((size_t)steps, 0);
        auto collect = [&](int s) -> int { // rank 0: the batch of step s has arrived (every rank's block), while later steps are in flight
            CHECK(orbfe_pipeline_gathered_wait(p, sets[(size_t)s]));
            for (int r = 0; r < world; r++) {
                uint8_t* d_block = nullptr;
                CHECK(orbfe_pipeline_gathered_set(p, sets[(size_t)s], r, &d_block));
                CHECK((int)hipMemcpy(rec.data(), d_block, rec.size(), hipMemcpyDeviceToHost));
                if (write_file(prefix + ".step" + std::to_string(s) + ".rank" + std::to_string(r), rec.data(), rec.size())) return 5;
            }
            CHECK(orbfe_pipeline_gathered_release(p, sets[(size_t)s], nullptr));
            return 0;
        };
        for (int s = 0; s < steps; s++) {
            CHECK(orbfe_pipeline_step(p, d_frames + (size_t)(s % nb) * batch_bytes, pitch, &sets[(size_t)s]));
            if (every_step) {
                CHECK(orbfe_pipeline_synchronize(p));
                uint8_t* d_rec = nullptr;
                CHECK(orbfe_pipeline_records(p, sets[(size_t)s], &d_rec));
                CHECK((int)hipMemcpy(rec.data(), d_rec, rec.size(), hipMemcpyDeviceToHost));
                if (write_file(prefix + ".own.step" + std::to_string(s), rec.data(), rec.size())) return 5;
            } else if (rank == 0 && s >= 1 && collect(s - 1))
                return 1;
        }
        CHECK(orbfe_pipeline_synchronize(p));
        if (!every_step && rank == 0 && collect(steps - 1)) return 1;
        int32_t st[4];
        CHECK(orbfe_pipeline_status(p, st));
        if (st[0] || st[1] || st[2] || st[3]) { fprintf(stderr, "capacity flags %d %d %d %d\n", st[0], st[1], st[2], st[3]); return 4; }
        printf("ok rank %d of %d steps %d batches %d record_bytes %llu\n", rank, world, steps, nb, (unsigned long long)lay.nbytes);
        orbfe_pipeline_destroy(p);
        (void)hipFree(d_frames);
        return 0;
    }

    const bool force_gather = argc > 7 && atoi(argv[7]);
    if (force_gather) { // the gather branch on the one GPU that is there: a communicator of one rank
        uint8_t id[128];
        CHECK(orbfe_pipeline_comm_unique_id(id));
        CHECK(orbfe_pipeline_comm_init(p, id, 0, 1, 0));
    }
    int32_t set = 0;
    for (int s = 0; s < steps; s++) CHECK(orbfe_pipeline_step(p, d_frames + (size_t)(s % nb) * batch_bytes, pitch, &set));   // the stream loops
    CHECK(orbfe_pipeline_synchronize(p));
    int32_t st[4];
    CHECK(orbfe_pipeline_status(p, st));
    if (st[0] || st[1] || st[2] || st[3]) { fprintf(stderr, "capacity flags %d %d %d %d\n", st[0], st[1], st[2], st[3]); return 4; }
    uint8_t* d_rec = nullptr;
    if (force_gather) CHECK(orbfe_pipeline_gathered(p, 0, &d_rec));   // what the destination rank received
    else CHECK(orbfe_pipeline_records(p, set, &d_rec));
    CHECK((int)hipMemcpy(rec.data(), d_rec, rec.size(), hipMemcpyDeviceToHost));
    int32_t* d_nm = nullptr;
    CHECK(orbfe_pipeline_matches(p, nullptr, nullptr, nullptr, nullptr, &d_nm));
    std::vector<int32_t> nm((size_t)B);
    CHECK((int)hipMemcpy(nm.data(), d_nm, nm.size() * 4, hipMemcpyDeviceToHost));
    if (write_file(argv[6], rec.data(), rec.size(), nm.data(), nm.size() * 4)) return 5;
    long nk = 0, nmk = 0, nmatch = 0;
    const int32_t* n = reinterpret_cast<const int32_t*>(rec.data() + lay.off_n) + lay.halo;
    const int32_t* m = reinterpret_cast<const int32_t*>(rec.data() + lay.off_nmarkers);
    for (int i = 0; i < B; i++) { nk += n[i]; nmk += m[i]; nmatch += nm[(size_t)i]; }
    printf("ok frames %d keypoints %ld markers %ld matches %ld record_bytes %llu\n", B, nk, nmk, nmatch, (unsigned long long)lay.nbytes);
    orbfe_pipeline_destroy(p);
    (void)hipFree(d_frames);
    return 0;
}
