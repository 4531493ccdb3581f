/* ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the keyframe feature records of the reference's binary map files,
 * for tests/.  Never linked into or called by the product library.  PARITY UNPINNED (the reference ships no map file).
 *
 * What it follows: src/Map.cc:297-321 (Map::SaveKeyFrame, the per-feature loop: one f.write per field, in this order) and
 * src/Map.cc:478-511 (Map::LoadKeyFrame, the matching f.read sequence). */
#include <climits>
#include <cstdint>
#include <cstring>

namespace {
struct KeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };

struct Writer {
    uint8_t* p;
    template <class T> void write(const T& v) { std::memcpy(p, &v, sizeof(T)); p += sizeof(T); }
};
struct Reader {
    const uint8_t* p;
    template <class T> void read(T& v) { std::memcpy(&v, p, sizeof(T)); p += sizeof(T); }
};
} // namespace

extern "C" {

/* returns bytes written (68 per feature) */
long oracle_keyframe_features_pack(const void* kps_, const uint8_t* desc, const uint64_t* mp_index, int n, uint8_t* out)
{
    const KeyPoint* kps = (const KeyPoint*)kps_;
    Writer f{out};
    for (int i = 0; i < n; i++) {
        const KeyPoint& kp = kps[i];
        f.write(kp.x);
        f.write(kp.y);
        f.write(kp.size);
        f.write(kp.angle);
        f.write(kp.response);
        f.write(kp.octave);
        const int cols = 32; /* kf->mDescriptors.cols is always 32 here (Map.cc:309) */
        f.write(cols);
        for (int j = 0; j < cols; j++) f.write(desc[32 * (size_t)i + j]);
        unsigned long int mnIdx = mp_index ? (unsigned long int)mp_index[i] : ULONG_MAX;
        f.write(mnIdx);
    }
    return (long)(f.p - out);
}

/* returns the number of bytes consumed, or -(i + 1) when record i has a descriptor length other than 32 */
long oracle_keyframe_features_unpack(const uint8_t* in, int n, void* kps_, uint8_t* desc, uint64_t* mp_index)
{
    KeyPoint* kps = (KeyPoint*)kps_;
    Reader f{in};
    for (int i = 0; i < n; i++) {
        KeyPoint kp;
        kp.class_id = -1; /* cv::KeyPoint() */
        f.read(kp.x);
        f.read(kp.y);
        f.read(kp.size);
        f.read(kp.angle);
        f.read(kp.response);
        f.read(kp.octave);
        kps[i] = kp;
        int cols = 0;
        f.read(cols);
        if (cols != 32) return -(long)(i + 1);
        for (int j = 0; j < cols; j++) f.read(desc[32 * (size_t)i + j]);
        unsigned long int mpidx;
        f.read(mpidx);
        if (mp_index) mp_index[i] = (uint64_t)mpidx;
    }
    return (long)(f.p - in);
}

} /* extern "C" */
