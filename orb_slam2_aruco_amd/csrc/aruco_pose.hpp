// Marker pose on the device: the IPPE planar pose solver as Marker::calculateExtrinsics runs it (reference
// Thirdparty/aruco/aruco/marker.cpp:322-344 -> ippe.cpp:91-100 -> PoseSolver::solveGeneric, ippe.cpp:138-222) and as
// Frame.cc:170 calls aruco::solvePnP for both solutions and their reprojection errors (ippe.cpp:72-89).
// One lane per marker; everything is closed form in double except the 3x3 symmetric eigenproblem of the homography
// estimator (HomographyHO, ippe.cpp:899-1032), which is a cyclic Jacobi here as it is in cv::eigen.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/orbfe.h"

struct PoseCamera {
    double fx, fy, cx, cy;
    double k[12]; // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    int has_dist; // distortion vector not empty (cv::undistortPoints skips the iterations otherwise)
};

namespace pose {

struct V2 {
    double x, y;
};

// cv::undistortPoints without R and P: normalised coordinates, rounded to float like the CV_32FC2 result (ippe.cpp:149)
__device__ inline V2 normalise_point(float u, float v, const PoseCamera& c)
{
    double x = ((double)u - c.cx) * (1. / c.fx), y = ((double)v - c.cy) * (1. / c.fy);
    if (c.has_dist) {
        const double x0 = x, y0 = y;
        const double* k = c.k;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
    }
    return {(double)(float)x, (double)(float)y};
}

// isotropic normalisation of 4 points (ippe.cpp:803-897): returns beta and the mean; p is overwritten with the normalised points
__device__ inline void normalise4(V2 p[4], double& beta, double& xm, double& ym)
{
    xm = 0; ym = 0;
    for (int i = 0; i < 4; i++) { xm = xm + p[i].x; ym = ym + p[i].y; }
    xm = xm / 4.0; ym = ym / 4.0;
    double kappa = 0;
    for (int i = 0; i < 4; i++) {
        p[i].x -= xm; p[i].y -= ym;
        kappa = kappa + p[i].x * p[i].x + p[i].y * p[i].y;
    }
    beta = sqrt(8 / kappa);
    for (int i = 0; i < 4; i++) { p[i].x *= beta; p[i].y *= beta; }
}

// eigenvector of the smallest eigenvalue of a symmetric 3x3 (cyclic Jacobi, rows of E are the eigenvectors)
__device__ inline void smallest_eigenvector(double a00, double a01, double a02, double a11, double a12, double a22, double h[3])
{
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double E[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 30; sweep++) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 2.2250738585072014e-308) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2, r = 3 - p - q;
            const double apq = A[p][q];
            if (fabs(apq) < 2.2250738585072014e-308) continue;
            const double y = (A[q][q] - A[p][p]) * 0.5;
            double t = fabs(y) + hypot(apq, y);
            double s = hypot(apq, t);
            const double c = t / s;
            s = apq / s;
            t = (apq / t) * apq;
            if (y < 0) { s = -s; t = -t; }
            A[p][p] -= t;
            A[q][q] += t;
            A[p][q] = A[q][p] = 0;
            const double arp = A[r][p], arq = A[r][q];
            A[r][p] = A[p][r] = arp * c - arq * s;
            A[r][q] = A[q][r] = arp * s + arq * c;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double ep = E[p][i], eq = E[q][i];
                E[p][i] = ep * c - eq * s;
                E[q][i] = ep * s + eq * c;
            }
        }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    // cv::eigen sorts descending and the estimator takes the last row; among equal eigenvalues the later one stays last
    if (m == 0 && A[1][1] == A[0][0]) m = 1;
    if (m <= 1 && A[2][2] == A[m][m]) m = 2;
    h[0] = m == 0 ? E[0][0] : m == 1 ? E[1][0] : E[2][0];
    h[1] = m == 0 ? E[0][1] : m == 1 ? E[1][1] : E[2][1];
    h[2] = m == 0 ? E[0][2] : m == 1 ? E[1][2] : E[2][2];
}

// HomographyHO::homographyHO (ippe.cpp:899-1032) for four correspondences a -> b, H(2,2) = 1
__device__ inline void homography4(const V2 src[4], const V2 targ[4], double H[9])
{
    V2 a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = src[i]; b[i] = targ[i]; }
    double betaA, xmA, ymA, betaB, xmB, ymB;
    normalise4(a, betaA, xmA, ymA);
    normalise4(b, betaB, xmB, ymB);
    double c1[4], c2[4], c3[4], c4[4], m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    for (int i = 0; i < 4; i++) {
        c1[i] = -b[i].x * a[i].x; c2[i] = -b[i].x * a[i].y; c3[i] = -b[i].y * a[i].x; c4[i] = -b[i].y * a[i].y;
        m1 = m1 + c1[i]; m2 = m2 + c2[i]; m3 = m3 + c3[i]; m4 = m4 + c4[i];
    }
    m1 = m1 / 4; m2 = m2 / 4; m3 = m3 / 4; m4 = m4 / 4;
    double Mx[4][3], My[4][3];
    for (int i = 0; i < 4; i++) {
        Mx[i][0] = c1[i] - m1; Mx[i][1] = c2[i] - m2; Mx[i][2] = -b[i].x;
        My[i][0] = c3[i] - m3; My[i][1] = c4[i] - m4; My[i][2] = -b[i].y;
    }
    double g00 = 0, g01 = 0, g11 = 0;
    for (int i = 0; i < 4; i++) { g00 += a[i].x * a[i].x; g01 += a[i].x * a[i].y; g11 += a[i].y * a[i].y; }
    const double dt = g00 * g11 - g01 * g01;
    const double i00 = g11 / dt, i01 = -g01 / dt, i11 = g00 / dt;
    double Bx[2][3] = {{0, 0, 0}, {0, 0, 0}}, By[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < 4; i++) {
        const double p0 = i00 * a[i].x + i01 * a[i].y, p1 = i01 * a[i].x + i11 * a[i].y;
        for (int j = 0; j < 3; j++) {
            Bx[0][j] += p0 * Mx[i][j]; Bx[1][j] += p1 * Mx[i][j];
            By[0][j] += p0 * My[i][j]; By[1][j] += p1 * My[i][j];
        }
    }
    double s[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; // D^T D, rows of D = Mx - A^T Bx then My - A^T By, summed in that order
    for (int half = 0; half < 2; half++)
        for (int i = 0; i < 4; i++) {
            double d[3];
            for (int j = 0; j < 3; j++)
                d[j] = half == 0 ? Mx[i][j] - (a[i].x * Bx[0][j] + a[i].y * Bx[1][j]) : My[i][j] - (a[i].x * By[0][j] + a[i].y * By[1][j]);
            for (int u = 0; u < 3; u++)
                for (int v = u; v < 3; v++) s[u][v] += d[u] * d[v];
        }
    double h[3];
    smallest_eigenvector(s[0][0], s[0][1], s[0][2], s[1][1], s[1][2], s[2][2], h);
    double n[9];
    n[0] = -(Bx[0][0] * h[0] + Bx[0][1] * h[1] + Bx[0][2] * h[2]);
    n[1] = -(Bx[1][0] * h[0] + Bx[1][1] * h[1] + Bx[1][2] * h[2]);
    n[2] = -(m1 * h[0] + m2 * h[1]);
    n[3] = -(By[0][0] * h[0] + By[0][1] * h[1] + By[0][2] * h[2]);
    n[4] = -(By[1][0] * h[0] + By[1][1] * h[1] + By[1][2] * h[2]);
    n[5] = -(m3 * h[0] + m4 * h[1]);
    n[6] = h[0]; n[7] = h[1]; n[8] = h[2];
    // H = TB * n * TAi with TB = [1/bB 0 xmB; 0 1/bB ymB; 0 0 1], TAi = [bA 0 -bA xmA; 0 bA -bA ymA; 0 0 1]
    const double ib = 1.0 / betaB;
    double t[9];
    for (int j = 0; j < 3; j++) {
        t[j] = ib * n[j] + 0 * n[3 + j] + xmB * n[6 + j];
        t[3 + j] = 0 * n[j] + ib * n[3 + j] + ymB * n[6 + j];
        t[6 + j] = 0 * n[j] + 0 * n[3 + j] + 1 * n[6 + j];
    }
    const double tx = -betaA * xmA, ty = -betaA * ymA;
    for (int i = 0; i < 3; i++) {
        H[3 * i] = t[3 * i] * betaA + t[3 * i + 1] * 0 + t[3 * i + 2] * 0;
        H[3 * i + 1] = t[3 * i] * 0 + t[3 * i + 1] * betaA + t[3 * i + 2] * 0;
        H[3 * i + 2] = t[3 * i] * tx + t[3 * i + 1] * ty + t[3 * i + 2] * 1;
    }
    const double h22 = H[8];
    for (int i = 0; i < 9; i++) H[i] = H[i] / h22;
}

// the two rotations from the homography's Jacobian at the origin (ippe.cpp:482-583 with :1035-1077 inlined)
__device__ inline void rotations(double j00, double j01, double j10, double j11, double p, double q, double R1[9], double R2[9])
{
    double ax = p, ay = q, az = 1;
    const double nrm = sqrt(ax * ax + ay * ay + az * az);
    ax = ax / nrm; ay = ay / nrm; az = az / nrm;
    double Ra[9];
    if (fabs(1.0 + az) < 1.1920928955078125e-07) {
        for (int i = 0; i < 9; i++) Ra[i] = 0;
        Ra[0] = 1.0; Ra[4] = 1.0; Ra[8] = -1.0;
    } else {
        const double d = 1.0 / (1.0 + az), ax2 = ax * ax, ay2 = ay * ay, axay = ax * ay;
        Ra[0] = -ax2 * d + 1.0; Ra[1] = -axay * d; Ra[2] = -ax;
        Ra[3] = -axay * d; Ra[4] = -ay2 * d + 1.0; Ra[5] = -ay;
        Ra[6] = ax; Ra[7] = ay; Ra[8] = 1.0 - (ax2 + ay2) * d;
    }
    // Rv = Ra^T
    const double rv[3][3] = {{Ra[0], Ra[3], Ra[6]}, {Ra[1], Ra[4], Ra[7]}, {Ra[2], Ra[5], Ra[8]}};
    const double b00 = rv[0][0] - p * rv[2][0], b01 = rv[0][1] - p * rv[2][1], b10 = rv[1][0] - q * rv[2][0], b11 = rv[1][1] - q * rv[2][1];
    const double dtinv = 1.0 / ((b00 * b11 - b01 * b10));
    const double bi00 = dtinv * b11, bi01 = -dtinv * b01, bi10 = -dtinv * b10, bi11 = dtinv * b00;
    const double a00 = bi00 * j00 + bi01 * j10, a01 = bi00 * j01 + bi01 * j11, a10 = bi10 * j00 + bi11 * j10, a11 = bi10 * j01 + bi11 * j11;
    const double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    const double gamma = sqrt(0.5 * (ata00 + ata11 + sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01)));
    const double r00 = a00 / gamma, r01 = a01 / gamma, r10 = a10 / gamma, r11 = a11 / gamma;
    const double b0 = sqrt(-r00 * r00 - r10 * r10 + 1);
    double b1 = sqrt(-r01 * r01 - r11 * r11 + 1);
    if ((-r00 * r01 - r10 * r11) < 0) b1 = -b1;
    for (int i = 0; i < 3; i++) {
        const double v0 = rv[i][0], v1 = rv[i][1], v2 = rv[i][2];
        R1[3 * i] = (r00)*v0 + (r10)*v1 + (b0)*v2;
        R1[3 * i + 1] = (r01)*v0 + (r11)*v1 + (b1)*v2;
        R1[3 * i + 2] = (b1 * r10 - b0 * r11) * v0 + (b0 * r01 - b1 * r00) * v1 + (r00 * r11 - r01 * r10) * v2;
        R2[3 * i] = (r00)*v0 + (r10)*v1 + (-b0) * v2;
        R2[3 * i + 1] = (r01)*v0 + (r11)*v1 + (-b1) * v2;
        R2[3 * i + 2] = (b0 * r11 - b1 * r10) * v0 + (b1 * r00 - b0 * r01) * v1 + (r00 * r11 - r01 * r10) * v2;
    }
}

// least-squares translation for a rotation (ippe.cpp:399-480), n = 4
__device__ inline void translation(const V2 obj[4], const V2 img[4], const double R[9], double t[3])
{
    double A02 = 0, A12 = 0, A22 = 0, b0 = 0, b1 = 0, b2 = 0;
    const double A00 = 4, A11 = 4;
    for (int i = 0; i < 4; i++) {
        const double rx = R[0] * obj[i].x + R[1] * obj[i].y, ry = R[3] * obj[i].x + R[4] * obj[i].y, rz = R[6] * obj[i].x + R[7] * obj[i].y;
        const double a2 = -img[i].x, c2 = -img[i].y;
        A02 = A02 + a2; A12 = A12 + c2;
        A22 = A22 + a2 * a2 + c2 * c2;
        const double bx = -a2 * rz - rx, by = -c2 * rz - ry;
        b0 = b0 + bx; b1 = b1 + by; b2 = b2 + a2 * bx + c2 * by;
    }
    const double A20 = A02, A21 = A12;
    const double detAInv = 1.0 / (A00 * A11 * A22 - A00 * A12 * A21 - A02 * A11 * A20);
    const double S00 = A11 * A22 - A12 * A21, S01 = A02 * A21, S02 = -A02 * A11;
    const double S10 = A12 * A20, S11 = A00 * A22 - A02 * A20, S12 = -A00 * A12;
    const double S20 = -A11 * A20, S21 = -A00 * A21, S22 = A00 * A11;
    t[0] = detAInv * (S00 * b0 + S01 * b1 + S02 * b2);
    t[1] = detAInv * (S10 * b0 + S11 * b1 + S12 * b2);
    t[2] = detAInv * (S20 * b0 + S21 * b1 + S22 * b2);
}

// PoseSolver::rot2vec (ippe.cpp:368-397)
__device__ inline void rot2vec(const double R[9], double r[3])
{
    const double w = acos((R[0] + R[4] + R[8] - 1.0) / 2.0);
    const double d = 1 / (2 * sin(w)) * w;
    if (w < 1.1920928955078125e-07) {
        r[0] = r[1] = r[2] = 0;
    } else {
        r[0] = d * (R[7] - R[5]); r[1] = d * (R[2] - R[6]); r[2] = d * (R[3] - R[1]);
    }
}

// evalReprojError (ippe.cpp:746-779): rot2vec -> cv::projectPoints (Rodrigues, distortion, float result) -> float RMS
__device__ inline float reprojection_error(const float obj[4][2], const float img[4][2], const PoseCamera& c, const double Rm[9],
                                           const double t[3])
{
    double r[3], R[9];
    rot2vec(Rm, r);
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < 2.220446049250313e-16) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        const double cs = cos(theta), sn = sin(theta), c1 = 1. - cs, it = 1. / theta;
        const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
        R[0] = cs + c1 * (x * x) + sn * 0; R[1] = cs * 0 + c1 * (x * y) + sn * -z; R[2] = cs * 0 + c1 * (x * z) + sn * y;
        R[3] = cs * 0 + c1 * (x * y) + sn * z; R[4] = cs + c1 * (y * y) + sn * 0; R[5] = cs * 0 + c1 * (y * z) + sn * -x;
        R[6] = cs * 0 + c1 * (x * z) + sn * -y; R[7] = cs * 0 + c1 * (y * z) + sn * x; R[8] = cs + c1 * (z * z) + sn * 0;
    }
    const double* k = c.k;
    float err = 0;
    for (int i = 0; i < 4; i++) {
        const double X = obj[i][0], Y = obj[i][1], Z = 0.0;
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z != 0 ? 1. / z : 1;
        x *= z; y *= z;
        const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        const double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        const double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4;
        const double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
        const float dx = (float)(xd * c.fx + c.cx) - img[i][0], dy = (float)(yd * c.fy + c.cy) - img[i][1];
        err += dx * dx + dy * dy;
    }
    return sqrtf(err / (2.0f * 4));
}

// aruco::solvePnP(get3DPoints(size), corners, K, dist): both poses, sorted by reprojection error
__device__ inline void solve_marker(const float corners[4][2], float marker_size, const PoseCamera& c, orbfe_marker_pose* out)
{
    const float hs = marker_size / 2.f; // marker.cpp:360
    const float obj[4][2] = {{-hs, hs}, {hs, hs}, {hs, -hs}, {-hs, -hs}};
    V2 img[4], can[4];
    double xb = 0, yb = 0;
    for (int i = 0; i < 4; i++) {
        img[i] = normalise_point(corners[i][0], corners[i][1], c);
        xb += (double)obj[i][0]; yb += (double)obj[i][1];
    }
    xb = xb / 4.0; yb = yb / 4.0;
    for (int i = 0; i < 4; i++) can[i] = {(double)obj[i][0] - xb, (double)obj[i][1] - yb};
    double H[9];
    homography4(can, img, H);
    const double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2], j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    double Ra[9], Rb[9], ta[3], tb[3];
    rotations(j00, j01, j10, j11, H[2], H[5], Ra, Rb);
    translation(can, img, Ra, ta);
    translation(can, img, Rb, tb);
    for (int i = 0; i < 3; i++) { // the canonical frame is the centred one (MCenter, ippe.cpp:707-722)
        ta[i] = Ra[3 * i] * -xb + Ra[3 * i + 1] * -yb + ta[i];
        tb[i] = Rb[3 * i] * -xb + Rb[3 * i + 1] * -yb + tb[i];
    }
    const float ea = reprojection_error(obj, corners, c, Ra, ta), eb = reprojection_error(obj, corners, c, Rb, tb);
    const bool a_first = ea < eb; // ippe.cpp:786
    double r1[3], r2[3];
    rot2vec(a_first ? Ra : Rb, r1);
    rot2vec(a_first ? Rb : Ra, r2);
    for (int i = 0; i < 3; i++) {
        out->rvec[i] = (float)r1[i];
        out->tvec[i] = (float)(a_first ? ta[i] : tb[i]);
        out->rvec2[i] = (float)r2[i];
        out->tvec2[i] = (float)(a_first ? tb[i] : ta[i]);
    }
    out->err[0] = a_first ? ea : eb;
    out->err[1] = a_first ? eb : ea;
}

// One of the two solutions (which = 0: a, 1: b) of solve_marker(), for a kernel that gives a marker two lanes: the shared part -- the
// normalised corners, the homography, both rotations -- then this solution's translation, reprojection error and rotation vector.
// The same operations on the same values as solve_marker() performs for that solution.
__device__ inline void solve_marker_half(const float corners[4][2], float marker_size, const PoseCamera& c, int which, float* err, float rvec[3],
                                         float tvec[3])
{
    const float hs = marker_size / 2.f; // marker.cpp:360
    const float obj[4][2] = {{-hs, hs}, {hs, hs}, {hs, -hs}, {-hs, -hs}};
    V2 img[4], can[4];
    double xb = 0, yb = 0;
    for (int i = 0; i < 4; i++) {
        img[i] = normalise_point(corners[i][0], corners[i][1], c);
        xb += (double)obj[i][0]; yb += (double)obj[i][1];
    }
    xb = xb / 4.0; yb = yb / 4.0;
    for (int i = 0; i < 4; i++) can[i] = {(double)obj[i][0] - xb, (double)obj[i][1] - yb};
    double H[9];
    homography4(can, img, H);
    const double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2], j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    double Ra[9], Rb[9], R[9], t[3];
    rotations(j00, j01, j10, j11, H[2], H[5], Ra, Rb);
    for (int i = 0; i < 9; i++) R[i] = which ? Rb[i] : Ra[i];
    translation(can, img, R, t);
    for (int i = 0; i < 3; i++) t[i] = R[3 * i] * -xb + R[3 * i + 1] * -yb + t[i]; // the canonical frame is the centred one (MCenter, ippe.cpp:707-722)
    *err = reprojection_error(obj, corners, c, R, t);
    double r[3];
    rot2vec(R, r);
    for (int i = 0; i < 3; i++) { rvec[i] = (float)r[i]; tvec[i] = (float)t[i]; }
}

} // namespace pose
