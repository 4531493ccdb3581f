/* ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the marker pose step of the reference, for tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline.  Never linked into or called by the product library.
 * PARITY UNPINNED: the reference has no tests or golden vectors and OpenCV (undistortPoints, projectPoints, Rodrigues,
 * eigen) is not in this image; their OpenCV 3.4 algorithms are restated from the published source.
 *
 * What it follows (reference tree Thirdparty/aruco/aruco):
 *   marker.cpp:322-344      Marker::calculateExtrinsics -> get3DPoints (:358-369) -> aruco::solvePnP (ippe.cpp:91-100)
 *   ippe.cpp:72-89          aruco::solvePnP returning both poses and their reprojection errors (Frame.cc:170-174)
 *   ippe.cpp:138-168        PoseSolver::solveGeneric (outer): undistort, solve, sort, rot2vec
 *   ippe.cpp:170-222        solveGeneric (inner): canonical points, homographyHO, solveCanonicalForm
 *   ippe.cpp:224-266        solveCanonicalForm
 *   ippe.cpp:368-397        rot2vec
 *   ippe.cpp:399-480        computeTranslation
 *   ippe.cpp:482-583        computeRotations, :1035-1077 rotateVec2ZAxis
 *   ippe.cpp:647-744        makeCanonicalObjectPoints (z = 0 branch: the marker's own corners)
 *   ippe.cpp:746-779        evalReprojError, :781-801 sortPosesByReprojError
 *   ippe.cpp:803-897        HomographyHO::normalizeDataIsotropic, :899-1032 homographyHO
 *   cameraparameters.cpp:158-173  CameraParameters::resize (markerdetector_impl.cpp:1110-1172 calls it when the image
 *                                 size differs from CamSize)
 * Plain double arithmetic in the reference's order; matrices are small fixed arrays. */
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace {

struct Cam {
    double fx, fy, cx, cy, k[14];
    int ndist;
};

Cam make_cam(const float* K4, const float* dist, int ndist)
{
    Cam c;
    c.fx = K4[0]; c.fy = K4[1]; c.cx = K4[2]; c.cy = K4[3];
    for (int i = 0; i < 14; i++) c.k[i] = i < ndist ? (double)dist[i] : 0.0;
    c.ndist = ndist;
    return c;
}

/* cv::undistortPoints(src, dst, K, dist) with no R and no P: normalised coordinates, stored as float (CV_32FC2). */
void undistort_normalized(const float* src, int n, const Cam& c, float* dst)
{
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    for (int i = 0; i < n; i++) {
        double x = src[2 * i], y = src[2 * i + 1];
        x = (x - c.cx) * ifx;
        y = (y - c.cy) * ify;
        if (c.ndist > 0) {
            const double* k = c.k;
            double x0 = x, y0 = y;
            for (int j = 0; j < 5; j++) {
                double r2 = x * x + y * y;
                double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
                double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
                double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
                x = (x0 - deltaX) * icdist;
                y = (y0 - deltaY) * icdist;
            }
        }
        dst[2 * i] = (float)x;
        dst[2 * i + 1] = (float)y;
    }
}

/* cv::Rodrigues, vector -> matrix (calib3d, double path). */
void rodrigues(const double r[3], double R[9])
{
    double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int i = 0; i < 9; i++) R[i] = c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i] + s * r_x[i];
}

/* cv::projectPoints for float object points: double arithmetic, float result (the output takes the object points' depth). */
void project_points(const float* obj, int n, const double rvec[3], const double t[3], const Cam& c, float* out)
{
    double R[9];
    rodrigues(rvec, R);
    const double* k = c.k;
    for (int i = 0; i < n; i++) {
        double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4;
        double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
        out[2 * i] = (float)(xd * c.fx + c.cx);
        out[2 * i + 1] = (float)(yd * c.fy + c.cy);
    }
}

/* cv::eigen of a symmetric 3x3 (Jacobi): eigenvalues descending, eigenvectors as rows. */
void eigen_sym3(const double Ain[9], double w[3], double V[9])
{
    double A[3][3], E[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A[i][j] = Ain[3 * i + j];
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
        if (off < DBL_MIN) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (std::fabs(A[p][q]) < DBL_MIN) continue;
                double y = (A[q][q] - A[p][p]) * 0.5;
                double t = std::fabs(y) + std::hypot(A[p][q], y);
                double s = std::hypot(A[p][q], t);
                double c = t / s;
                s = A[p][q] / s;
                t = (A[p][q] / t) * A[p][q];
                if (y < 0) { s = -s; t = -t; }
                /* rotation in the (p, q) plane zeroing A[p][q] */
                double App = A[p][p] - t, Aqq = A[q][q] + t;
                int r = 3 - p - q;
                double Arp = A[r][p], Arq = A[r][q];
                A[p][p] = App; A[q][q] = Aqq; A[p][q] = A[q][p] = 0;
                A[r][p] = A[p][r] = Arp * c - Arq * s;
                A[r][q] = A[q][r] = Arp * s + Arq * c;
                for (int i = 0; i < 3; i++) {
                    double ep = E[p][i], eq = E[q][i];
                    E[p][i] = ep * c - eq * s;
                    E[q][i] = ep * s + eq * c;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (A[idx[j]][idx[j]] > A[idx[i]][idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    for (int i = 0; i < 3; i++) {
        w[i] = A[idx[i]][idx[i]];
        for (int j = 0; j < 3; j++) V[3 * i + j] = E[idx[i]][j];
    }
}

/* HomographyHO::normalizeDataIsotropic for 4 points given as (x, y) doubles (ippe.cpp:803-897). */
void normalize_isotropic(const double* data, int n, double* DataN /*2 x n*/, double T[9], double Ti[9])
{
    double xm = 0, ym = 0;
    for (int i = 0; i < n; i++) { xm = xm + data[2 * i]; ym = ym + data[2 * i + 1]; }
    xm = xm / (double)n;
    ym = ym / (double)n;
    double kappa = 0;
    for (int i = 0; i < n; i++) {
        double xh = data[2 * i] - xm, yh = data[2 * i + 1] - ym;
        DataN[i] = xh;
        DataN[n + i] = yh;
        kappa = kappa + xh * xh + yh * yh;
    }
    double beta = std::sqrt(2 * n / kappa);
    for (int i = 0; i < 2 * n; i++) DataN[i] = DataN[i] * beta;
    std::memset(T, 0, 9 * sizeof(double));
    std::memset(Ti, 0, 9 * sizeof(double));
    T[0] = 1.0 / beta; T[4] = 1.0 / beta; T[2] = xm; T[5] = ym; T[8] = 1;
    Ti[0] = beta; Ti[4] = beta; Ti[2] = -beta * xm; Ti[5] = -beta * ym; Ti[8] = 1;
}

void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

/* HomographyHO::homographyHO (ippe.cpp:899-1032) for n = 4. */
void homography_ho(const double* src, const double* targ, double H[9])
{
    const int n = 4;
    double DataA[2 * n], DataB[2 * n], TA[9], TAi[9], TB[9], TBi[9];
    normalize_isotropic(src, n, DataA, TA, TAi);
    normalize_isotropic(targ, n, DataB, TB, TBi);
    double C1[n], C2[n], C3[n], C4[n], Mx[n][3], My[n][3];
    double mC1 = 0, mC2 = 0, mC3 = 0, mC4 = 0;
    for (int i = 0; i < n; i++) {
        C1[i] = -DataB[i] * DataA[i];
        C2[i] = -DataB[i] * DataA[n + i];
        C3[i] = -DataB[n + i] * DataA[i];
        C4[i] = -DataB[n + i] * DataA[n + i];
        mC1 = mC1 + C1[i]; mC2 = mC2 + C2[i]; mC3 = mC3 + C3[i]; mC4 = mC4 + C4[i];
    }
    mC1 = mC1 / n; mC2 = mC2 / n; mC3 = mC3 / n; mC4 = mC4 / n;
    for (int i = 0; i < n; i++) {
        Mx[i][0] = C1[i] - mC1; Mx[i][1] = C2[i] - mC2; Mx[i][2] = -DataB[i];
        My[i][0] = C3[i] - mC3; My[i][1] = C4[i] - mC4; My[i][2] = -DataB[n + i];
    }
    double G[2][2] = {{0, 0}, {0, 0}}; /* DataA * DataA^T */
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int i = 0; i < n; i++) G[a][b] += DataA[a * n + i] * DataA[b * n + i];
    double dt = G[0][0] * G[1][1] - G[0][1] * G[1][0];
    double Gi[2][2] = {{G[1][1] / dt, -G[0][1] / dt}, {-G[1][0] / dt, G[0][0] / dt}};
    double Pp[2][n]; /* Gi * DataA */
    for (int a = 0; a < 2; a++)
        for (int i = 0; i < n; i++) Pp[a][i] = Gi[a][0] * DataA[i] + Gi[a][1] * DataA[n + i];
    double Bx[2][3], By[2][3];
    for (int a = 0; a < 2; a++)
        for (int j = 0; j < 3; j++) {
            double sx = 0, sy = 0;
            for (int i = 0; i < n; i++) { sx += Pp[a][i] * Mx[i][j]; sy += Pp[a][i] * My[i][j]; }
            Bx[a][j] = sx; By[a][j] = sy;
        }
    double D[2 * n][3];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
            double ex = DataA[i] * Bx[0][j] + DataA[n + i] * Bx[1][j]; /* (DataA^T * Bx)(i, j) */
            double ey = DataA[i] * By[0][j] + DataA[n + i] * By[1][j];
            D[i][j] = Mx[i][j] - ex;
            D[i + n][j] = My[i][j] - ey;
        }
    double DDT[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double s = 0;
            for (int i = 0; i < 2 * n; i++) s += D[i][a] * D[i][b];
            DDT[3 * a + b] = s;
        }
    double S[3], U[9];
    eigen_sym3(DDT, S, U);
    const double h7 = U[6], h8 = U[7], h9 = U[8];
    double Hn[9];
    Hn[0] = -(Bx[0][0] * h7 + Bx[0][1] * h8 + Bx[0][2] * h9);
    Hn[1] = -(Bx[1][0] * h7 + Bx[1][1] * h8 + Bx[1][2] * h9);
    Hn[2] = -(mC1 * h7 + mC2 * h8);
    Hn[3] = -(By[0][0] * h7 + By[0][1] * h8 + By[0][2] * h9);
    Hn[4] = -(By[1][0] * h7 + By[1][1] * h8 + By[1][2] * h9);
    Hn[5] = -(mC3 * h7 + mC4 * h8);
    Hn[6] = h7; Hn[7] = h8; Hn[8] = h9;
    double tmp[9];
    mat3_mul(TB, Hn, tmp);
    mat3_mul(tmp, TAi, H);
    const double h22 = H[8];
    for (int i = 0; i < 9; i++) H[i] = H[i] / h22;
}

/* PoseSolver::rotateVec2ZAxis (ippe.cpp:1035-1077) */
void rotate_vec_to_z(const double a[3], double Ra[9])
{
    double ax = a[0], ay = a[1], az = a[2];
    double nrm = std::sqrt(ax * ax + ay * ay + az * az);
    ax = ax / nrm; ay = ay / nrm; az = az / nrm;
    double c = az;
    if (std::fabs(1.0 + c) < std::numeric_limits<float>::epsilon()) {
        std::memset(Ra, 0, 9 * sizeof(double));
        Ra[0] = 1.0; Ra[4] = 1.0; Ra[8] = -1.0;
    } else {
        double d = 1.0 / (1.0 + c), ax2 = ax * ax, ay2 = ay * ay, axay = ax * ay;
        Ra[0] = -ax2 * d + 1.0; Ra[1] = -axay * d; Ra[2] = -ax;
        Ra[3] = -axay * d; Ra[4] = -ay2 * d + 1.0; Ra[5] = -ay;
        Ra[6] = ax; Ra[7] = ay; Ra[8] = 1.0 - (ax2 + ay2) * d;
    }
}

/* PoseSolver::computeRotations (ippe.cpp:482-583) */
void compute_rotations(double j00, double j01, double j10, double j11, double p, double q, double R1[9], double R2[9])
{
    double v[3] = {p, q, 1}, Rz[9], Rv[9];
    rotate_vec_to_z(v, Rz);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rv[3 * i + j] = Rz[3 * j + i];
    const double rv00 = Rv[0], rv01 = Rv[1], rv02 = Rv[2], rv10 = Rv[3], rv11 = Rv[4], rv12 = Rv[5], rv20 = Rv[6], rv21 = Rv[7],
                 rv22 = Rv[8];
    double b00 = rv00 - p * rv20, b01 = rv01 - p * rv21, b10 = rv10 - q * rv20, b11 = rv11 - q * rv21;
    double dtinv = 1.0 / ((b00 * b11 - b01 * b10));
    double binv00 = dtinv * b11, binv01 = -dtinv * b01, binv10 = -dtinv * b10, binv11 = dtinv * b00;
    double a00 = binv00 * j00 + binv01 * j10, a01 = binv00 * j01 + binv01 * j11;
    double a10 = binv10 * j00 + binv11 * j10, a11 = binv10 * j01 + binv11 * j11;
    double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    double gamma = std::sqrt(0.5 * (ata00 + ata11 + std::sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01)));
    double rt00 = a00 / gamma, rt01 = a01 / gamma, rt10 = a10 / gamma, rt11 = a11 / gamma;
    double b0 = std::sqrt(-rt00 * rt00 - rt10 * rt10 + 1), b1 = std::sqrt(-rt01 * rt01 - rt11 * rt11 + 1);
    double sp = (-rt00 * rt01 - rt10 * rt11);
    if (sp < 0) b1 = -b1;
    const double rvr[3][3] = {{rv00, rv01, rv02}, {rv10, rv11, rv12}, {rv20, rv21, rv22}};
    for (int i = 0; i < 3; i++) {
        const double r0 = rvr[i][0], r1 = rvr[i][1], r2 = rvr[i][2];
        R1[3 * i + 0] = (rt00)*r0 + (rt10)*r1 + (b0)*r2;
        R1[3 * i + 1] = (rt01)*r0 + (rt11)*r1 + (b1)*r2;
        R1[3 * i + 2] = (b1 * rt10 - b0 * rt11) * r0 + (b0 * rt01 - b1 * rt00) * r1 + (rt00 * rt11 - rt01 * rt10) * r2;
        R2[3 * i + 0] = (rt00)*r0 + (rt10)*r1 + (-b0) * r2;
        R2[3 * i + 1] = (rt01)*r0 + (rt11)*r1 + (-b1) * r2;
        R2[3 * i + 2] = (b0 * rt11 - b1 * rt10) * r0 + (b1 * rt00 - b0 * rt01) * r1 + (rt00 * rt11 - rt01 * rt10) * r2;
    }
}

/* PoseSolver::computeTranslation (ippe.cpp:399-480) */
void compute_translation(const double* obj2 /*n x 2*/, const double* img /*n x 2*/, int n, const double R[9], double t[3])
{
    double ATA00 = n, ATA02 = 0, ATA11 = n, ATA12 = 0, ATA20 = 0, ATA21 = 0, ATA22 = 0, ATb0 = 0, ATb1 = 0, ATb2 = 0;
    for (int i = 0; i < n; i++) {
        double rx = R[0] * obj2[2 * i] + R[1] * obj2[2 * i + 1];
        double ry = R[3] * obj2[2 * i] + R[4] * obj2[2 * i + 1];
        double rz = R[6] * obj2[2 * i] + R[7] * obj2[2 * i + 1];
        double a2 = -img[2 * i], b2 = -img[2 * i + 1];
        ATA02 = ATA02 + a2; ATA12 = ATA12 + b2; ATA20 = ATA20 + a2; ATA21 = ATA21 + b2;
        ATA22 = ATA22 + a2 * a2 + b2 * b2;
        double bx = -a2 * rz - rx, by = -b2 * rz - ry;
        ATb0 = ATb0 + bx; ATb1 = ATb1 + by; ATb2 = ATb2 + a2 * bx + b2 * by;
    }
    double detAInv = 1.0 / (ATA00 * ATA11 * ATA22 - ATA00 * ATA12 * ATA21 - ATA02 * ATA11 * ATA20);
    double S00 = ATA11 * ATA22 - ATA12 * ATA21, S01 = ATA02 * ATA21, S02 = -ATA02 * ATA11;
    double S10 = ATA12 * ATA20, S11 = ATA00 * ATA22 - ATA02 * ATA20, S12 = -ATA00 * ATA12;
    double S20 = -ATA11 * ATA20, S21 = -ATA00 * ATA21, S22 = ATA00 * ATA11;
    t[0] = detAInv * (S00 * ATb0 + S01 * ATb1 + S02 * ATb2);
    t[1] = detAInv * (S10 * ATb0 + S11 * ATb1 + S12 * ATb2);
    t[2] = detAInv * (S20 * ATb0 + S21 * ATb1 + S22 * ATb2);
}

/* PoseSolver::rot2vec (ippe.cpp:368-397) */
void rot2vec(const double R[9], double r[3])
{
    double trace = R[0] + R[4] + R[8];
    double w_norm = std::acos((trace - 1.0) / 2.0);
    double eps = std::numeric_limits<float>::epsilon();
    double d = 1 / (2 * std::sin(w_norm)) * w_norm;
    if (w_norm < eps) {
        r[0] = r[1] = r[2] = 0;
    } else {
        r[0] = d * (R[7] - R[5]);
        r[1] = d * (R[2] - R[6]);
        r[2] = d * (R[3] - R[1]);
    }
}

/* PoseSolver::evalReprojError (ippe.cpp:746-779): float differences, float accumulation */
float eval_reproj_error(const float* obj3, const float* img, int n, const Cam& c, const double R[9], const double t[3])
{
    double r[3];
    rot2vec(R, r);
    float proj[8];
    project_points(obj3, n, r, t, c, proj);
    float err = 0;
    for (int i = 0; i < n; i++) {
        float dx = proj[2 * i] - img[2 * i], dy = proj[2 * i + 1] - img[2 * i + 1];
        err += dx * dx + dy * dy;
    }
    return std::sqrt(err / (2.0f * n));
}

} // namespace

extern "C" {

/* aruco::solvePnP(get3DPoints(size), corners, K, dist) (ippe.cpp:72-100, marker.cpp:333-338): both IPPE poses of one square
 * marker, sorted by reprojection error.  out = rvec1[3] tvec1[3] rvec2[3] tvec2[3] (double), err = {err1, err2}. */
void oracle_marker_pose(const float* corners /*4 x (x, y)*/, float marker_size, const float* K4, const float* dist, int ndist,
                        double* out, float* err)
{
    const Cam c = make_cam(K4, dist, ndist);
    const float halfSize = marker_size / 2.f; /* marker.cpp:360 */
    const float obj3[12] = {-halfSize, halfSize, 0, halfSize, halfSize, 0, halfSize, -halfSize, 0, -halfSize, -halfSize, 0};
    float nrm_f[8];
    undistort_normalized(corners, 4, c, nrm_f); /* ippe.cpp:149 */
    double nrm[8];
    for (int i = 0; i < 8; i++) nrm[i] = nrm_f[i];
    /* makeCanonicalObjectPoints, float object points (z is not examined for CV_32FC3): centre, keep (x, y) */
    double xBar = 0, yBar = 0, zBar = 0, U[12];
    for (int i = 0; i < 4; i++) {
        U[3 * i] = obj3[3 * i]; U[3 * i + 1] = obj3[3 * i + 1]; U[3 * i + 2] = obj3[3 * i + 2];
        xBar += U[3 * i]; yBar += U[3 * i + 1]; zBar += U[3 * i + 2];
    }
    xBar = xBar / 4.0; yBar = yBar / 4.0; zBar = zBar / 4.0;
    double canon[8];
    for (int i = 0; i < 4; i++) { canon[2 * i] = U[3 * i] - xBar; canon[2 * i + 1] = U[3 * i + 1] - yBar; }
    double H[9];
    homography_ho(canon, nrm, H);
    /* solveCanonicalForm */
    double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2], j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    double v0 = H[2], v1 = H[5];
    double Ra[9], Rb[9], ta[3], tb[3];
    compute_rotations(j00, j01, j10, j11, v0, v1, Ra, Rb);
    compute_translation(canon, nrm, 4, Ra, ta);
    compute_translation(canon, nrm, 4, Rb, tb);
    /* Ma = MaCanon * MCenter: t += R * (-bar) */
    for (int i = 0; i < 3; i++) {
        ta[i] = Ra[3 * i] * -xBar + Ra[3 * i + 1] * -yBar + Ra[3 * i + 2] * -zBar + ta[i];
        tb[i] = Rb[3 * i] * -xBar + Rb[3 * i + 1] * -yBar + Rb[3 * i + 2] * -zBar + tb[i];
    }
    float erra = eval_reproj_error(obj3, corners, 4, c, Ra, ta), errb = eval_reproj_error(obj3, corners, 4, c, Rb, tb);
    const double *R1 = Ra, *t1 = ta, *R2 = Rb, *t2 = tb;
    if (erra < errb) {
        err[0] = erra; err[1] = errb;
    } else {
        err[0] = errb; err[1] = erra;
        R1 = Rb; t1 = tb; R2 = Ra; t2 = ta;
    }
    rot2vec(R1, out);
    std::memcpy(out + 3, t1, 3 * sizeof(double));
    rot2vec(R2, out + 6);
    std::memcpy(out + 9, t2, 3 * sizeof(double));
}

/* CameraParameters::resize (cameraparameters.cpp:158-173): float factors applied to the float camera matrix. */
void oracle_camera_resize(const float* K4, int cam_w, int cam_h, int img_w, int img_h, float* out)
{
    std::memcpy(out, K4, 4 * sizeof(float));
    if (img_w == cam_w && img_h == cam_h) return;
    float AxFactor = float(img_w) / float(cam_w), AyFactor = float(img_h) / float(cam_h);
    out[0] *= AxFactor; out[2] *= AxFactor; out[1] *= AyFactor; out[3] *= AyFactor;
}

} /* extern "C" */
