// The batched-video mode from C++, without Python or PyTorch: what an application that links liborbfe.so writes (include/orbfe.h,
// "the batched-video mode").  TEST INFRASTRUCTURE (tests/test_pipeline_gpu.py compiles and runs it and compares the record set it
// writes with the Python wrapper's, byte for byte):
//     pipeline_driver <frames.u8> <frames> <rows> <cols> <steps> <out.records> [force_gather]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/orbfe.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s: %d (%s)\n", #call, rc_, orbfe_last_error()); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 7) return 2;
    const int B = atoi(argv[2]), rows = atoi(argv[3]), cols = atoi(argv[4]), steps = atoi(argv[5]);
    const size_t pitch = ((size_t)cols + 63) / 64 * 64;
    std::vector<uint8_t> host((size_t)B * rows * cols), padded((size_t)B * rows * pitch, 0);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(host.data(), 1, host.size(), f) != host.size()) return 3;
    fclose(f);
    for (size_t r = 0; r < (size_t)B * rows; r++) std::copy(host.begin() + r * cols, host.begin() + (r + 1) * cols, padded.begin() + r * pitch);
    uint8_t* d_frames = nullptr;
    CHECK((int)hipMalloc(&d_frames, padded.size()));
    CHECK((int)hipMemcpy(d_frames, padded.data(), padded.size(), hipMemcpyHostToDevice));

    orbfe_pipeline_config cfg;
    CHECK(orbfe_pipeline_config_default(&cfg, B, rows, cols));
    orbfe_pipeline* p = orbfe_pipeline_create(&cfg);
    if (!p) { fprintf(stderr, "orbfe_pipeline_create: %s\n", orbfe_last_error()); return 1; }
    if (argc > 7 && atoi(argv[7])) { // the gather branch on the one GPU that is there: a communicator of one rank
        uint8_t id[128];
        CHECK(orbfe_pipeline_comm_unique_id(id));
        CHECK(orbfe_pipeline_comm_init(p, id, 0, 1, 0));
    }
    int32_t set = 0;
    for (int s = 0; s < steps; s++) CHECK(orbfe_pipeline_step(p, d_frames, pitch, &set));   // the same batch again: a looping stream
    CHECK(orbfe_pipeline_synchronize(p));
    int32_t st[4];
    CHECK(orbfe_pipeline_status(p, st));
    if (st[0] || st[1] || st[2] || st[3]) { fprintf(stderr, "capacity flags %d %d %d %d\n", st[0], st[1], st[2], st[3]); return 4; }
    orbfe_record_layout lay;
    CHECK(orbfe_pipeline_layout(p, &lay));
    uint8_t* d_rec = nullptr;
    if (argc > 7 && atoi(argv[7])) CHECK(orbfe_pipeline_gathered(p, 0, &d_rec));   // what the destination rank received
    else CHECK(orbfe_pipeline_records(p, set, &d_rec));
    std::vector<uint8_t> rec(lay.nbytes);
    CHECK((int)hipMemcpy(rec.data(), d_rec, rec.size(), hipMemcpyDeviceToHost));
    int32_t* d_nm = nullptr;
    CHECK(orbfe_pipeline_matches(p, nullptr, nullptr, nullptr, nullptr, &d_nm));
    std::vector<int32_t> nm((size_t)B);
    CHECK((int)hipMemcpy(nm.data(), d_nm, nm.size() * 4, hipMemcpyDeviceToHost));
    f = fopen(argv[6], "wb");
    if (!f) return 5;
    fwrite(rec.data(), 1, rec.size(), f);
    fwrite(nm.data(), 4, nm.size(), f);
    fclose(f);
    long nk = 0, nmk = 0, nmatch = 0;
    const int32_t* n = reinterpret_cast<const int32_t*>(rec.data() + lay.off_n) + lay.halo;
    const int32_t* m = reinterpret_cast<const int32_t*>(rec.data() + lay.off_nmarkers);
    for (int i = 0; i < B; i++) { nk += n[i]; nmk += m[i]; nmatch += nm[(size_t)i]; }
    printf("ok frames %d keypoints %ld markers %ld matches %ld record_bytes %llu\n", B, nk, nmk, nmatch, (unsigned long long)lay.nbytes);
    orbfe_pipeline_destroy(p);
    (void)hipFree(d_frames);
    return 0;
}
