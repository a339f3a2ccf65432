// srl_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code; see srl_oracle.h).
//
// Line-by-line restatement of the SR-LIVO LIO scan-matching hot path.  Every function cites
// the reference file:line (relative to /root/reference) it follows.  The reference has no tests for this
// path; the restatement is pinned BITWISE against the reference's own translation units compiled in place
// (oracle/_ref/libref_path.so, tests/test_reference_tu.py) -- see the header of srl_oracle.h.
//
// Third-party arithmetic that is NOT under /root/reference (Eigen 3.3.x, unpinned; libstdc++)
// is restated from its published algorithms:
//   * Quaterniond::toRotationMatrix / Quaterniond(Matrix3d) (Shepperd) / normalized()
//   * Matrix<double,17,17>::inverse()  -> partial-pivot LU
//   * SelfAdjointEigenSolver<Matrix3d> -> eig3_eigen_ql: Eigen 3.3.7's own algorithm (scale by max |a_ij|, the 3x3
//     real specialisation of tridiagonalization_inplace, implicit symmetric QR steps with Wilkinson shift and
//     Givens rotations, selection sort ascending), operation by operation.  eig3_jacobi (FP64 cyclic Jacobi) is kept
//     as an independent second solver (orc_set_eig_solver) to bound the difference on every configuration.
//   * 3-vector reductions in the fixed order (x*x + y*y) + z*z, no FMA contraction
//   * std::priority_queue is used literally (libstdc++ heap), so tie order is the real one.
//
// Build: see oracle/Makefile.  -O3, no -march, no fast-math (matches CMakeLists.txt:4).
#include "srl_oracle.h"
#include "orc_eigen337.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <queue>
#include <tuple>
#include <unordered_map>
#include <random>
#include <tr1/unordered_map>
#include <vector>

#ifdef ORC_USE_TSL
#include <tsl/robin_map.h>   // the real vendored header: thirdLibrary/tessil-src/include
#endif

namespace {
using orc_algos::lu_inverse;
using orc_algos::eig3_eigen_ql;

// ---------------------------------------------------------------------------------------------
// mini linear algebra (plain structs; row-major)
// ---------------------------------------------------------------------------------------------
struct V3 { double x, y, z; };
struct M3 { double m[3][3]; };
struct Q4 { double w, x, y, z; };

inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(const V3 &a, const V3 &b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3 &a, const V3 &b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(const V3 &a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(double s, const V3 &a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(const V3 &a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
inline double dot(const V3 &a, const V3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline double sqnorm(const V3 &a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
inline double norm(const V3 &a) { return std::sqrt(sqnorm(a)); }
inline V3 cross(const V3 &a, const V3 &b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Eigen 3.3 normalize(): divide by sqrt(squaredNorm) when squaredNorm > 0
inline V3 normalized(const V3 &a) {
    double z = sqnorm(a);
    if (z > 0.0) return a / std::sqrt(z);
    return a;
}
inline M3 m3_identity() { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = (i == j) ? 1.0 : 0.0; return r; }
inline M3 m3_zero() { M3 r; std::memset(&r, 0, sizeof r); return r; }
inline M3 m3_from(const double *p) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = p[3 * i + j]; return r; }
inline void m3_to(const M3 &a, double *p) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) p[3 * i + j] = a.m[i][j]; }
// Eigen column-wise evaluation: res = (col0*v0 + col1*v1) + col2*v2
inline V3 operator*(const M3 &a, const V3 &v) {
    return v3((a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z,
              (a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z,
              (a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z);
}
inline M3 operator*(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
    return r;
}
inline M3 operator*(const M3 &a, double s) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] * s; return r; }
inline M3 operator+(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j]; return r; }
inline M3 operator-(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j]; return r; }
inline M3 transpose(const M3 &a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i]; return r; }

// numType::skewSymmetric  (include/utility.h:204-212)
inline M3 skew(const V3 &v) {
    M3 r;
    r.m[0][0] = 0;    r.m[0][1] = -v.z; r.m[0][2] = v.y;
    r.m[1][0] = v.z;  r.m[1][1] = 0;    r.m[1][2] = -v.x;
    r.m[2][0] = -v.y; r.m[2][1] = v.x;  r.m[2][2] = 0;
    return r;
}

// Eigen::Quaterniond::toRotationMatrix -- no normalisation inside
inline M3 quat_to_rot(const Q4 &q) {
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r.m[0][0] = 1.0 - (tyy + tzz); r.m[0][1] = txy - twz;         r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;         r.m[1][1] = 1.0 - (txx + tzz); r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;         r.m[2][1] = tyz + twx;         r.m[2][2] = 1.0 - (txx + tyy);
    return r;
}
// Eigen::Quaterniond(Matrix3d): Shepperd's method
inline Q4 rot_to_quat(const M3 &a) {
    Q4 q;
    double t = a.m[0][0] + a.m[1][1] + a.m[2][2];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (a.m[2][1] - a.m[1][2]) * t;
        q.y = (a.m[0][2] - a.m[2][0]) * t;
        q.z = (a.m[1][0] - a.m[0][1]) * t;
    } else {
        int i = 0;
        if (a.m[1][1] > a.m[0][0]) i = 1;
        if (a.m[2][2] > a.m[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0);
        double qv[3];
        qv[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (a.m[k][j] - a.m[j][k]) * t;
        qv[j] = (a.m[j][i] + a.m[i][j]) * t;
        qv[k] = (a.m[k][i] + a.m[i][k]) * t;
        q.x = qv[0]; q.y = qv[1]; q.z = qv[2];
    }
    return q;
}
inline double q_sqnorm(const Q4 &q) { return ((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w; }
inline Q4 q_normalized(const Q4 &q) {
    double z = q_sqnorm(q);
    if (z > 0.0) { double n = std::sqrt(z); return Q4{q.w / n, q.x / n, q.y / n, q.z / n}; }
    return q;
}
inline Q4 q_mul(const Q4 &a, const Q4 &b) {
    return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
              a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
// Eigen::Quaterniond::inverse(): conjugate / squaredNorm
inline Q4 q_inverse(const Q4 &q) {
    double n2 = q_sqnorm(q);
    if (n2 > 0.0) return Q4{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
    return Q4{0, 0, 0, 0};
}

// Eigen::QuaternionBase::_transformVector (Quaternion * Vector3, Eigen 3.3 Geometry/Quaternion.h): NOT via a matrix
inline V3 q_rotate(const Q4 &q, const V3 &v) {
    const V3 qv = v3(q.x, q.y, q.z);
    V3 uv = cross(qv, v);
    uv = uv + uv;
    return (v + q.w * uv) + cross(qv, uv);
}

// numType::normalizeR (utility.h:194-202)
inline M3 normalizeR(const M3 &R) { return quat_to_rot(q_normalized(rot_to_quat(R))); }

#define ORC_THETA_THRESHOLD 0.0001   // utility.h:27

// numType::rotationToSo3 (utility.h:266-278)
inline V3 rotation_to_so3(const M3 &R_in) {
    M3 R = normalizeR(R_in);
    double theta = std::acos((R.m[0][0] + R.m[1][1] + R.m[2][2] - 1.0) / 2.0);
    V3 a = v3(R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]);
    if (theta < ORC_THETA_THRESHOLD) return a / 2.0;
    return (theta * a) / (2.0 * std::sin(theta));
}
// numType::so3ToRotation (utility.h:280-297)
inline M3 so3_to_rotation(const V3 &w) {
    double theta = norm(w);
    if (theta < ORC_THETA_THRESHOLD) {
        M3 u = skew(w);
        return m3_identity() + u + (u * u) * 0.5;   // I + u_x + 0.5*u_x*u_x (scalar first: 0.5*u_x, then *u_x)
    }
    M3 u = skew(normalized(w));
    return m3_identity() + u * std::sin(theta) + (u * (1.0 - std::cos(theta))) * u;
}
// numType::so3ToQuat (utility.h:299-324)
inline Q4 so3_to_quat(const V3 &w) {
    double theta = norm(w);
    if (theta < ORC_THETA_THRESHOLD) {
        V3 h = w / 2.0;
        return q_normalized(Q4{1.0, h.x, h.y, h.z});
    }
    V3 u = normalized(w);
    double s = std::sin(0.5 * theta);
    return q_normalized(Q4{std::cos(0.5 * theta), u.x * s, u.y * s, u.z * s});
}
// numType::quatToSo3 (utility.h:326-330)
inline V3 quat_to_so3(const Q4 &q) { return rotation_to_so3(quat_to_rot(q)); }

// numType::derivativeS2 (utility.h:214-233): 3x2, B[i][j]
struct M32 { double m[3][2]; };
inline M32 derivative_s2(const V3 &g_in) {
    V3 g = normalized(g_in);
    M32 B;
    B.m[0][0] = 1.0 - g.x * g.x / (1.0 + g.z);
    B.m[0][1] = -g.x * g.y / (1.0 + g.z);
    B.m[1][0] = B.m[0][1];
    B.m[1][1] = 1.0 - g.y * g.y / (1.0 + g.z);
    B.m[2][0] = -g.x;
    B.m[2][1] = -g.y;
    return B;
}
// AngularDistance(const Vector3d&) (src/utility.cpp:146-153), degrees
inline double angular_distance(const V3 &d_so3) {
    M3 dR = so3_to_rotation(d_so3);
    double n = ((dR.m[0][0] + dR.m[1][1] + dR.m[2][2]) - 1.0) / 2.0;
    return std::acos(n) * 180.0 / M_PI;
}

// lu_inverse<N> (Matrix<double,17,17>::inverse()) and eig3_eigen_ql (SelfAdjointEigenSolver<Matrix3d>): orc_eigen337.h
// SelfAdjointEigenSolver<Matrix3d> restated as FP64 cyclic Jacobi (eigenvalues ascending,
// eigenvectors = columns of V, unit norm, sign arbitrary).  Same routine is used (re-typed)
// by the HIP kernel, so the two agree to rounding.
void eig3_jacobi(const double Ain[3][3], double evals[3], double V[3][3]) {
    double a[3][3];
    double scale = 0.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) scale = std::max(scale, std::fabs(Ain[i][j]));
    if (!(scale > 0.0)) scale = 1.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { a[i][j] = Ain[i][j] / scale; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 32; sweep++) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-40) break;
        for (int p = 0; p < 2; p++) {
            for (int q = p + 1; q < 3; q++) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                double c = 1.0 / std::sqrt(t * t + 1.0);
                double s = t * c;
                int r = 3 - p - q;
                double app = a[p][p], aqq = a[q][q];
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
                for (int k = 0; k < 3; k++) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
        }
    }
    double d[3] = {a[0][0] * scale, a[1][1] * scale, a[2][2] * scale};
    int idx[3] = {0, 1, 2};
    // stable ascending sort of three values
    if (d[idx[1]] < d[idx[0]]) std::swap(idx[0], idx[1]);
    if (d[idx[2]] < d[idx[1]]) std::swap(idx[1], idx[2]);
    if (d[idx[1]] < d[idx[0]]) std::swap(idx[0], idx[1]);
    double Vs[3][3];
    for (int c = 0; c < 3; c++) { evals[c] = d[idx[c]]; for (int k = 0; k < 3; k++) Vs[k][c] = V[k][idx[c]]; }
    std::memcpy(V, Vs, sizeof Vs);
}

// which solver compute_neighborhood uses: 0 = eig3_eigen_ql (what the reference executes), 1 = eig3_jacobi
static int g_eig_solver = 0;
// threads of the keypoint loop: 1 = the reference's single-threaded loop; > 1 = the all-cores CPU baseline (OpenMP)
static int g_threads = 1;

// ---------------------------------------------------------------------------------------------
// map types (include/cloudMap.h)
// ---------------------------------------------------------------------------------------------
struct voxel {                                  // cloudMap.h:124-145
    short x, y, z;
    voxel() = default;
    voxel(short x_, short y_, short z_) : x(x_), y(y_), z(z_) {}
    bool operator==(const voxel &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct voxel_hash {                             // cloudMap.h:173-184 (shorts sign-extend to size_t)
    std::size_t operator()(const voxel &vox) const {
        const size_t kP1 = 73856093;
        const size_t kP2 = 19349669;
        const size_t kP3 = 83492791;
        return vox.x * kP1 + vox.y * kP2 + vox.z * kP3;
    }
};
// rgbPoint (cloudMap.h:51-86): only `position` (Vector3f) is on the path, the other members are
// kept as dead payload so the record is 80 bytes like the reference's (AoS stride matters for
// the CPU baseline's memory behaviour).
struct alignas(16) rgbPoint {
    float position[3];
    short rgb[3];
    float cov_rgb[3];
    double observe_distance;
    double last_observe_time;
    int point_index;
    short N_rgb;
    short is_out_lier_count;
    double image_velocity[2];
    explicit rgbPoint(const V3 &p) {            // cloudMap.cpp:5-9: position_.cast<float>()
        std::memset(this, 0, sizeof *this);
        position[0] = (float)p.x; position[1] = (float)p.y; position[2] = (float)p.z;
    }
    V3 getPosition() const {                    // cloudMap.cpp:26-29: position.cast<double>()
        return v3((double)position[0], (double)position[1], (double)position[2]);
    }
};
static_assert(sizeof(rgbPoint) == 80, "rgbPoint stand-in must be 80 bytes (SURVEY Appendix D)");

struct voxelBlock {                             // cloudMap.h:147-169
    explicit voxelBlock(int num_points_ = 20) : num_points(num_points_) { points.reserve(num_points_); }
    std::vector<rgbPoint> points;
    double last_visited_time = 0.0;
    bool is_recent = false;
    int voxel_index = -1;                       // ORACLE ADDITION: creation order, defines point ids
    bool IsFull() const { return (size_t)num_points == points.size(); }
    void AddPoint(const rgbPoint &p) { points.push_back(p); }
    int NumPoints() const { return (int)points.size(); }
private:
    int num_points;
};

#ifdef ORC_USE_TSL
typedef tsl::robin_map<voxel, voxelBlock, voxel_hash> voxelHashMap;   // cloudMap.h:171
#define ORC_VALUE(it) ((it).value())
static const char *kBackend = "tsl::robin_map";
#else
typedef std::unordered_map<voxel, voxelBlock, voxel_hash> voxelHashMap;
#define ORC_VALUE(it) ((it)->second)
static const char *kBackend = "std::unordered_map";
#endif

}  // namespace

struct orc_map {
    voxelHashMap map;
    std::vector<voxel> creation_order;          // ORACLE ADDITION (export only)
};

namespace {

// lioOptimization::addPointToMap (src/lioOptimization.cpp:400-446)
bool add_point_to_map(orc_map *om, const rgbPoint &point, double voxel_size, int max_num_points_in_voxel,
                      double min_distance_points, int min_num_points) {
    voxelHashMap &map = om->map;
    V3 pos = point.getPosition();
    short kx = static_cast<short>(pos.x / voxel_size);
    short ky = static_cast<short>(pos.y / voxel_size);
    short kz = static_cast<short>(pos.z / voxel_size);

    auto search = map.find(voxel(kx, ky, kz));
    if (search != map.end()) {
        voxelBlock &voxel_block = ORC_VALUE(search);
        if (!voxel_block.IsFull()) {
            double sq_dist_min_to_points = 10 * voxel_size * voxel_size;
            for (int i = 0; i < voxel_block.NumPoints(); ++i) {
                const rgbPoint &_point = voxel_block.points[i];
                double sq_dist = sqnorm(_point.getPosition() - pos);
                if (sq_dist < sq_dist_min_to_points) sq_dist_min_to_points = sq_dist;
            }
            if (sq_dist_min_to_points > (min_distance_points * min_distance_points)) {
                if (min_num_points <= 0 || voxel_block.NumPoints() >= min_num_points) {
                    voxel_block.AddPoint(point);
                    return true;
                }
            }
        }
    } else {
        if (min_num_points <= 0) {
            voxelBlock voxel_block(max_num_points_in_voxel);
            voxel_block.AddPoint(point);
            voxel_block.voxel_index = (int)om->creation_order.size();
            om->creation_order.push_back(voxel(kx, ky, kz));
            map[voxel(kx, ky, kz)] = std::move(voxel_block);
            return true;
        }
    }
    return false;
}

// pair_distance_t / comparator / priority_queue_t (src/optimize.cpp:355-363), with the point id
// carried as an extra tuple member (does not take part in the comparison).
using pair_distance_t = std::tuple<double, V3, voxel, int32_t>;
struct comparator {
    bool operator()(const pair_distance_t &left, const pair_distance_t &right) const {
        return std::get<0>(left) < std::get<0>(right);
    }
};
using priority_queue_t = std::priority_queue<pair_distance_t, std::vector<pair_distance_t>, comparator>;

struct NeighborResult {
    std::vector<V3> pts;
    std::vector<int32_t> ids;
    std::vector<double> dist;
    bool tie = false;
    int num_candidates = 0;
};

// lioOptimization::searchNeighbors (src/optimize.cpp:365-426)
void search_neighbors(voxelHashMap &map, const V3 &point, int nb_voxels_visited, double size_voxel_map,
                      int max_num_neighbors, int threshold_voxel_capacity, int cap, NeighborResult &res,
                      bool want_tie) {
    short kx = static_cast<short>(point.x / size_voxel_map);
    short ky = static_cast<short>(point.y / size_voxel_map);
    short kz = static_cast<short>(point.z / size_voxel_map);

    priority_queue_t priority_queue;
    std::vector<double> all_dist;
    int num_candidates = 0;

    voxel voxel_temp(kx, ky, kz);
    for (short kxx = kx - nb_voxels_visited; kxx < kx + nb_voxels_visited + 1; ++kxx) {
        for (short kyy = ky - nb_voxels_visited; kyy < ky + nb_voxels_visited + 1; ++kyy) {
            for (short kzz = kz - nb_voxels_visited; kzz < kz + nb_voxels_visited + 1; ++kzz) {
                voxel_temp.x = kxx;
                voxel_temp.y = kyy;
                voxel_temp.z = kzz;

                auto search = map.find(voxel_temp);
                if (search != map.end()) {
                    voxelBlock &voxel_block = ORC_VALUE(search);
                    if (voxel_block.NumPoints() < threshold_voxel_capacity) continue;
                    for (int i = 0; i < voxel_block.NumPoints(); ++i) {
                        const rgbPoint &neighbor = voxel_block.points[i];
                        V3 neighbor_point = neighbor.getPosition();
                        double distance = norm(neighbor_point - point);
                        num_candidates++;
                        if (want_tie) all_dist.push_back(distance);
                        int32_t id = voxel_block.voxel_index * cap + i;
                        if ((int)priority_queue.size() == max_num_neighbors) {
                            if (distance < std::get<0>(priority_queue.top())) {
                                priority_queue.pop();
                                priority_queue.emplace(distance, neighbor_point, voxel_temp, id);
                            }
                        } else {
                            priority_queue.emplace(distance, neighbor_point, voxel_temp, id);
                        }
                    }
                }
            }
        }
    }

    auto size = priority_queue.size();
    res.pts.assign(size, v3(0, 0, 0));
    res.ids.assign(size, -1);
    res.dist.assign(size, 0.0);
    for (size_t i = 0; i < size; ++i) {
        res.pts[size - 1 - i] = std::get<1>(priority_queue.top());
        res.ids[size - 1 - i] = std::get<3>(priority_queue.top());
        res.dist[size - 1 - i] = std::get<0>(priority_queue.top());
        priority_queue.pop();
    }
    res.num_candidates = num_candidates;
    res.tie = false;
    if (want_tie && !all_dist.empty()) {
        // ORACLE ADDITION: exact tie inside the K+1 smallest distances => heap-order dependent result
        size_t m = std::min(all_dist.size(), (size_t)max_num_neighbors + 1);
        std::partial_sort(all_dist.begin(), all_dist.begin() + m, all_dist.end());
        for (size_t i = 1; i < m; i++) if (all_dist[i] == all_dist[i - 1]) { res.tie = true; break; }
    }
}

struct Neighborhood {                           // include/lioOptimization.h:127-137
    V3 center, normal;
    M3 covariance;
    double a2D;
    double evals[3];
};

// lioOptimization::computeNeighborhoodDistribution (src/optimize.cpp:316-353); returns false on NaN a2D
bool compute_neighborhood(const std::vector<V3> &points, Neighborhood &nb) {
    V3 barycenter = v3(0, 0, 0);
    for (const V3 &p : points) barycenter = barycenter + p;
    barycenter = barycenter / (double)points.size();
    nb.center = barycenter;

    M3 cov = m3_zero();
    for (const V3 &p : points) {
        const double pk[3] = {p.x, p.y, p.z};
        const double bk[3] = {barycenter.x, barycenter.y, barycenter.z};
        for (int k = 0; k < 3; ++k)
            for (int l = k; l < 3; ++l)
                cov.m[k][l] += (pk[k] - bk[k]) * (pk[l] - bk[l]);
    }
    cov.m[1][0] = cov.m[0][1];
    cov.m[2][0] = cov.m[0][2];
    cov.m[2][1] = cov.m[1][2];
    nb.covariance = cov;

    double V[3][3];
    if (g_eig_solver == 1) eig3_jacobi(cov.m, nb.evals, V);
    else eig3_eigen_ql(cov.m, nb.evals, V);          // Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> es(covariance_Matrix)
    nb.normal = normalized(v3(V[0][0], V[1][0], V[2][0]));   // es.eigenvectors().col(0).normalized()

    double sigma_1 = std::sqrt(std::abs(nb.evals[2]));
    double sigma_2 = std::sqrt(std::abs(nb.evals[1]));
    double sigma_3 = std::sqrt(std::abs(nb.evals[0]));
    nb.a2D = (sigma_2 - sigma_3) / sigma_1;
    return !(nb.a2D != nb.a2D);
}

struct planeParam {                             // cloudMap.h:97-108
    V3 raw_point, norm_vector;
    double jacobians[6];
    double norm_offset;
    double distance = 0.0;
    double weight = 1.0;
};

struct FrameCtx {
    Q4 rotation;        // p_frame->p_state->rotation
    V3 translation;     // p_frame->p_state->translation
    V3 last_translation;// all_cloud_frame[id-1]->p_state->translation (optimize.cpp:25)
    M3 R_imu_lidar;
    V3 t_imu_lidar;
    int frame_id;
};

// lioOptimization::buildPlaneResiduals (src/optimize.cpp:18-131)
// returns 0 ok / -2 NaN planarity; summary in neq.
int build_plane_residuals(orc_map *om, const orc_icp_opts &o, const double *raw_xyz, int n, const FrameCtx &f,
                          int cap, std::vector<planeParam> &plane_residuals, double &loss_sum,
                          orc_residual_out *out, orc_normal_eq *neq) {
    const short nb_voxels_visited = f.frame_id < o.init_num_frames ? 2 : (short)o.voxel_neighborhood;
    const int kMinNumNeighbors = o.min_number_neighbors;
    const int kThresholdCapacity = f.frame_id < o.init_num_frames ? 1 : o.threshold_voxel_occupancy;

    const Q4 end_quat = f.rotation;
    const V3 end_t = f.translation;

    double lambda_weight = std::abs(o.weight_alpha);
    double lambda_neighborhood = std::abs(o.weight_neighborhood);
    const double kMaxPointToPlane = o.max_dist_to_plane_icp;
    const double sum = lambda_weight + lambda_neighborhood;
    lambda_weight /= sum;
    lambda_neighborhood /= sum;

    int num_residuals = 0;
    const int num_keypoints = n;
    const int K = o.max_number_neighbors;

    // transformKeypoints (optimize.cpp:30-40)
    std::vector<V3> kp_point(n);
    {
        M3 R = quat_to_rot(q_normalized(end_quat));
        V3 t = end_t;
        for (int k = 0; k < n; k++) {
            V3 raw = v3(raw_xyz[3 * k], raw_xyz[3 * k + 1], raw_xyz[3 * k + 2]);
            kp_point[k] = R * (f.R_imu_lidar * raw + f.t_imu_lidar) + t;
        }
    }
    if (out) {
        if (out->status) std::memset(out->status, 3, (size_t)n);
        if (out->ids) for (size_t i = 0; i < (size_t)n * K; i++) out->ids[i] = -1;
        if (out->tie) std::memset(out->tie, 0, (size_t)n);
        if (out->point_world) for (int k = 0; k < n; k++) { out->point_world[3 * k] = kp_point[k].x; out->point_world[3 * k + 1] = kp_point[k].y; out->point_world[3 * k + 2] = kp_point[k].z; }
    }
    int64_t sum_candidates = 0;
    int num_visited = 0, num_ties = 0;
    const bool want_tie = (out && out->tie) || neq;

    // The loop body of optimize.cpp:68-108 in two halves: `visit` = everything one keypoint computes on its own
    // (searchNeighbors, estimatePointNeighborhood, weight, plane, distance, Jacobian); `commit` = what the sequential
    // loop does with it (counters, push_back, the break at :107).  Single-threaded (the reference; g_threads == 1) they
    // alternate keypoint by keypoint.  The all-cores CPU baseline (g_threads > 1, ORACLE ADDITION) visits a block of
    // keypoints in parallel and then commits the block in keypoint order -- same results bit for bit, because `visit`
    // has no shared state and `commit` still runs sequentially and stops where the reference stops.
    struct Visit {
        NeighborResult nres;
        bool enough = false, nan = false, accepted = false;
        Neighborhood neighborhood;
        planeParam plane;
    };
    const M3 R_end = quat_to_rot(end_quat);      // NOT normalised: end_quat.toRotationMatrix()
    auto visit = [&](int k, Visit &r) {
        V3 raw_point = v3(raw_xyz[3 * k], raw_xyz[3 * k + 1], raw_xyz[3 * k + 2]);
        search_neighbors(om->map, kp_point[k], nb_voxels_visited, o.size_voxel_map, o.max_number_neighbors,
                         kThresholdCapacity, cap, r.nres, want_tie);
        r.enough = !((int)r.nres.pts.size() < kMinNumNeighbors);
        r.nan = false; r.accepted = false;
        if (!r.enough) return;

        double weight;
        V3 location = f.R_imu_lidar * raw_point + f.t_imu_lidar;

        // estimatePointNeighborhood (optimize.cpp:42-53)
        if (!compute_neighborhood(r.nres.pts, r.neighborhood)) { r.nan = true; return; }   // optimize.cpp:348-350 throws
        weight = std::pow(r.neighborhood.a2D, o.power_planarity);
        if (dot(r.neighborhood.normal, f.last_translation - location) < 0) {
            r.neighborhood.normal = -1.0 * r.neighborhood.normal;
        }

        weight = lambda_weight * weight + lambda_neighborhood *
                 std::exp(-norm(r.nres.pts[0] - kp_point[k]) / (kMaxPointToPlane * kMinNumNeighbors));

        planeParam &plane_temp = r.plane;
        plane_temp.raw_point = location;
        plane_temp.norm_vector = normalized(r.neighborhood.normal);
        plane_temp.norm_offset = -dot(plane_temp.norm_vector, r.nres.pts[0]);
        plane_temp.distance = dot(plane_temp.norm_vector, R_end * plane_temp.raw_point + end_t) + plane_temp.norm_offset;
        plane_temp.weight = weight;

        if (plane_temp.distance < o.max_dist_to_plane_icp) {
            r.accepted = true;
            const V3 &nv = plane_temp.norm_vector;
            plane_temp.jacobians[0] = nv.x * weight;
            plane_temp.jacobians[1] = nv.y * weight;
            plane_temp.jacobians[2] = nv.z * weight;
            // - n^T * R * skew(raw_point) * weight, evaluated left to right
            const double mn[3] = {-nv.x, -nv.y, -nv.z};
            double r1[3];
            for (int j = 0; j < 3; j++) r1[j] = (mn[0] * R_end.m[0][j] + mn[1] * R_end.m[1][j]) + mn[2] * R_end.m[2][j];
            M3 S = skew(plane_temp.raw_point);
            for (int j = 0; j < 3; j++) {
                double r2 = (r1[0] * S.m[0][j] + r1[1] * S.m[1][j]) + r1[2] * S.m[2][j];
                plane_temp.jacobians[3 + j] = r2 * weight;
            }
        }
    };
    // returns 0 = go on, 1 = break (optimize.cpp:107), -2 = NaN planarity
    auto commit = [&](int k, const Visit &r) -> int {
        num_visited++;
        sum_candidates += r.nres.num_candidates;
        if (r.nres.tie) num_ties++;
        if (out) {
            if (out->tie) out->tie[k] = r.nres.tie ? 1 : 0;
            if (out->ids) for (size_t i = 0; i < r.nres.ids.size(); i++) out->ids[(size_t)k * K + i] = r.nres.ids[i];
            if (out->status) out->status[k] = 0;
        }
        if (!r.enough) return 0;                                  // `continue` (optimize.cpp:78-79)
        if (r.nan) { if (neq) neq->nan_error = 1; return -2; }
        const planeParam &plane_temp = r.plane;
        if (out) {
            if (out->status) out->status[k] = 1;
            if (out->normal) { out->normal[3 * k] = plane_temp.norm_vector.x; out->normal[3 * k + 1] = plane_temp.norm_vector.y; out->normal[3 * k + 2] = plane_temp.norm_vector.z; }
            if (out->a2D) out->a2D[k] = r.neighborhood.a2D;
            if (out->weight) out->weight[k] = plane_temp.weight;
            if (out->norm_offset) out->norm_offset[k] = plane_temp.norm_offset;
            if (out->distance) out->distance[k] = plane_temp.distance;
        }
        if (r.accepted) {
            num_residuals++;
            plane_residuals.push_back(plane_temp);
            loss_sum += plane_temp.distance * plane_temp.distance;
            if (out) {
                if (out->status) out->status[k] = 2;
                if (out->jacobian) for (int j = 0; j < 6; j++) out->jacobian[6 * k + j] = plane_temp.jacobians[j];
            }
        }
        if (num_residuals >= o.max_num_residuals) return 1;
        return 0;
    };

    if (g_threads <= 1) {
        Visit r;
        for (int k = 0; k < num_keypoints; k++) {
            visit(k, r);
            const int c = commit(k, r);
            if (c == -2) return -2;
            if (c == 1) break;
        }
    } else {
        // blocks sized like the device path's prefix pass, so that an early break wastes little parallel work
        const int max_res = o.max_num_residuals > 0 ? o.max_num_residuals : 1;
        const long long first = std::min<long long>(num_keypoints, 4LL * max_res + 2048);
        int k0 = 0;
        long long blk = first;
        std::vector<Visit> vis;
        bool stop = false;
        while (k0 < num_keypoints && !stop) {
            const int k1 = (int)std::min<long long>(num_keypoints, k0 + blk);
            vis.resize((size_t)(k1 - k0));
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads)
            for (int k = k0; k < k1; k++) visit(k, vis[(size_t)(k - k0)]);
            for (int k = k0; k < k1; k++) {
                const int c = commit(k, vis[(size_t)(k - k0)]);
                if (c == -2) return -2;
                if (c == 1) { stop = true; break; }
            }
            k0 = k1;
            blk = std::max<long long>(blk, 16384);
        }
    }

    if (neq) {
        neq->num_residuals = num_residuals;
        neq->success = (num_residuals < o.min_number_neighbors) ? 0 : 1;   // optimize.cpp:110
        neq->loss_sum = loss_sum;
        neq->sum_candidates = sum_candidates;
        neq->num_visited = num_visited;
        neq->num_ties = num_ties;
    }
    return 0;
}

// H_x^T H_x and H_x^T h (optimize.cpp:235,239) -- sequential over residuals
void normal_equations(const std::vector<planeParam> &pr, double HtH[36], double Hth[6]) {
    std::memset(HtH, 0, 36 * sizeof(double));
    std::memset(Hth, 0, 6 * sizeof(double));
    for (const planeParam &p : pr) {
        double h = p.distance * p.weight;          // optimize.cpp:169
        for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++) HtH[6 * a + b] += p.jacobians[a] * p.jacobians[b];
            Hth[a] += p.jacobians[a] * h;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// eskfEstimator (src/eskfEstimator.cpp)
// ---------------------------------------------------------------------------------------------
struct orc_eskf {
    V3 acc_0, gyr_0;
    V3 acc_cov, gyr_cov, b_acc_cov, b_gyr_cov;
    V3 p; Q4 q; V3 v, ba, bg, g;
    double noise[12][12];
    double covariance[17][17];
    // tryInit bookkeeping (eskfEstimator.h:26-45, utility.cpp:11,14)
    V3 acc_cov_scale, gyr_cov_scale, mean_gyr, mean_acc;
    bool is_first_imu_meas; double time_first_imu; int num_init_meas;
    bool initial_flag; double G_norm;
};

namespace {
void eskf_init(orc_eskf *e) {                   // eskfEstimator.cpp:3-21
    std::memset(e, 0, sizeof *e);
    for (int i = 0; i < 17; i++) e->covariance[i][i] = 1.0;
    e->p = v3(0, 0, 0);
    e->q = Q4{1, 0, 0, 0};
    e->v = e->ba = e->bg = v3(0, 0, 0);
    e->g = v3(0.0, 0.0, 9.81);
    e->mean_gyr = v3(0, 0, 0);
    e->mean_acc = v3(0, 0, 9.81);
    e->is_first_imu_meas = true;
    e->num_init_meas = 1;
    e->initial_flag = false;
    e->G_norm = 9.81;                           // lioOptimization.cpp:366-367 with the shipped G = (0,0,9.81)
}
inline V3 cwise_sq(const V3 &a) { return v3(a.x * a.x, a.y * a.y, a.z * a.z); }
void eskf_initialization(orc_eskf *e, const double *t, const double *gyr, const double *acc, int n) {   // eskfEstimator.cpp:93-118
    if (e->is_first_imu_meas) {
        e->num_init_meas = 1;
        e->is_first_imu_meas = false;
        e->time_first_imu = t[0];
        e->mean_gyr = v3(gyr[0], gyr[1], gyr[2]);
        e->mean_acc = v3(acc[0], acc[1], acc[2]);
    }
    for (int i = 0; i < n; i++) {
        const V3 w = v3(gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]), a = v3(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]);
        const int N = e->num_init_meas;         // int: "/ num_init_meas" divides by the converted double
        e->mean_gyr = e->mean_gyr + (w - e->mean_gyr) / (double)N;
        e->mean_acc = e->mean_acc + (a - e->mean_acc) / (double)N;
        // a * (N - 1.0) / N  parses as (a * (N - 1.0)) / N;  (N * N) is an int product
        e->gyr_cov = (e->gyr_cov * (N - 1.0)) / (double)N + (cwise_sq(w - e->mean_gyr) * (N - 1.0)) / (double)(N * N);
        e->acc_cov = (e->acc_cov * (N - 1.0)) / (double)N + (cwise_sq(a - e->mean_acc) * (N - 1.0)) / (double)(N * N);
        e->num_init_meas++;
    }
    e->gyr_0 = v3(gyr[3 * (n - 1)], gyr[3 * (n - 1) + 1], gyr[3 * (n - 1) + 2]);
    e->acc_0 = v3(acc[3 * (n - 1)], acc[3 * (n - 1) + 1], acc[3 * (n - 1) + 2]);
}
void eskf_observe(orc_eskf *e, const double d[17]) {   // eskfEstimator.cpp:219-230
    e->p = e->p + v3(d[0], d[1], d[2]);
    e->q = q_normalized(q_mul(e->q, so3_to_quat(v3(d[3], d[4], d[5]))));
    e->v = e->v + v3(d[6], d[7], d[8]);
    e->ba = e->ba + v3(d[9], d[10], d[11]);
    e->bg = e->bg + v3(d[12], d[13], d[14]);
    M32 B = derivative_s2(e->g);
    V3 so3_dg = v3(B.m[0][0] * d[15] + B.m[0][1] * d[16], B.m[1][0] * d[15] + B.m[1][1] * d[16],
                   B.m[2][0] * d[15] + B.m[2][1] * d[16]);
    e->g = so3_to_rotation(so3_dg) * e->g;
}
void eskf_predict(orc_eskf *e, double dt, const V3 &acc_1, const V3 &gyr_1) {   // eskfEstimator.cpp:166-217
    Q4 q_before = e->q;
    V3 un_gyr = 0.5 * (e->gyr_0 + gyr_1) - e->bg;
    V3 un_acc = 0.5 * (e->acc_0 + acc_1) - e->ba;
    e->q = q_mul(e->q, so3_to_quat(un_gyr * dt));
    e->p = e->p + e->v * dt;
    M3 Rb = quat_to_rot(q_before);
    e->v = e->v + (Rb * un_acc) * dt - e->g * dt;

    M3 R_omega_x = skew(un_gyr), R_acc_x = skew(un_acc);
    M32 B_x = derivative_s2(e->g);

    static double F_x[17][17], F_w[17][12];
    std::memset(F_x, 0, sizeof F_x);
    std::memset(F_w, 0, sizeof F_w);
    M3 I = m3_identity();
    auto set3 = [&](int r, int c, const M3 &m) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F_x[r + i][c + j] = m.m[i][j]; };
    set3(0, 0, I);
    set3(0, 6, I * dt);
    set3(3, 3, I - R_omega_x * dt);
    set3(3, 12, (I * -1.0) * dt);
    set3(6, 3, ((Rb * -1.0) * R_acc_x) * dt);
    set3(6, 6, I);
    set3(6, 9, (Rb * -1.0) * dt);
    M3 Sg = skew(e->g);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++)
            F_x[6 + i][15 + j] = ((Sg.m[i][0] * B_x.m[0][j] + Sg.m[i][1] * B_x.m[1][j]) + Sg.m[i][2] * B_x.m[2][j]) * dt;
    set3(9, 9, I);
    set3(12, 12, I);
    {
        double gn = norm(e->g);
        double f = -1.0 / (gn * gn);
        // -1/(|g|^2) * B^T * skew(g) * skew(g) * B, evaluated left to right like Eigen does
        double A1[2][3], A2[2][3], A3[2][3];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) A1[i][j] = f * B_x.m[j][i];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) A2[i][j] = (A1[i][0] * Sg.m[0][j] + A1[i][1] * Sg.m[1][j]) + A1[i][2] * Sg.m[2][j];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) A3[i][j] = (A2[i][0] * Sg.m[0][j] + A2[i][1] * Sg.m[1][j]) + A2[i][2] * Sg.m[2][j];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++)
            F_x[15 + i][15 + j] = (A3[i][0] * B_x.m[0][j] + A3[i][1] * B_x.m[1][j]) + A3[i][2] * B_x.m[2][j];
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        F_w[6 + i][0 + j] = -Rb.m[i][j] * dt;
        F_w[3 + i][3 + j] = -I.m[i][j] * dt;
        F_w[9 + i][6 + j] = -I.m[i][j] * dt;
        F_w[12 + i][9 + j] = -I.m[i][j] * dt;
    }
    static double T1[17][17], P1[17][17], T2[17][12], P2[17][17];
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) { double s = 0; for (int k = 0; k < 17; k++) s += F_x[i][k] * e->covariance[k][j]; T1[i][j] = s; }
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) { double s = 0; for (int k = 0; k < 17; k++) s += T1[i][k] * F_x[j][k]; P1[i][j] = s; }
    for (int i = 0; i < 17; i++) for (int j = 0; j < 12; j++) { double s = 0; for (int k = 0; k < 12; k++) s += F_w[i][k] * e->noise[k][j]; T2[i][j] = s; }
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) { double s = 0; for (int k = 0; k < 12; k++) s += T2[i][k] * F_w[j][k]; P2[i][j] = s; }
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) e->covariance[i][j] = P1[i][j] + P2[i][j];

    e->acc_0 = acc_1;
    e->gyr_0 = gyr_1;
}

// lioOptimization::updateIEKF (src/optimize.cpp:133-314)
int update_iekf(orc_map *om, orc_eskf *eskf_pro, const orc_icp_opts &o, const double *raw_xyz, int n,
                FrameCtx &f, V3 &velocity, V3 &ba_out, V3 &bg_out, int cap, double laser_point_cov,
                double *log, int max_log_iters, int *num_residuals_used) {
    int max_num_iter = f.frame_id < o.init_num_frames ? std::max(15, o.num_iters_icp) : o.num_iters_icp;

    V3 p_predict = eskf_pro->p;
    Q4 q_predict = eskf_pro->q;
    V3 v_predict = eskf_pro->v;
    V3 ba_predict = eskf_pro->ba;
    V3 bg_predict = eskf_pro->bg;
    V3 g_predict = eskf_pro->g;

    int iters = 0;
    if (num_residuals_used) *num_residuals_used = 0;

    for (int i = -1; i < max_num_iter; i++) {
        std::vector<planeParam> plane_residuals;
        double loss_old = 0.0;
        orc_normal_eq neq;
        std::memset(&neq, 0, sizeof neq);
        int rc = build_plane_residuals(om, o, raw_xyz, n, f, cap, plane_residuals, loss_old, nullptr, &neq);
        if (rc == -2) return -2;
        if (num_residuals_used) *num_residuals_used = neq.num_residuals;
        if (!neq.success) return -1;
        iters++;

        const int M = (int)plane_residuals.size();
        // H_x (M x 6), h (M)   (optimize.cpp:160-170)
        std::vector<double> H_x((size_t)M * 6), h(M);
        for (int r = 0; r < M; r++) {
            for (int c = 0; c < 6; c++) H_x[(size_t)r * 6 + c] = plane_residuals[r].jacobians[c];
            h[r] = plane_residuals[r].distance * plane_residuals[r].weight;
        }

        V3 d_p = eskf_pro->p - p_predict;
        Q4 d_q = q_mul(q_inverse(q_predict), eskf_pro->q);
        V3 d_so3 = quat_to_so3(d_q);
        V3 d_v = eskf_pro->v - v_predict;
        V3 d_ba = eskf_pro->ba - ba_predict;
        V3 d_bg = eskf_pro->bg - bg_predict;

        V3 g = eskf_pro->g;
        V3 g_predict_normalize = normalized(g_predict);
        V3 g_normalize = normalized(g);
        V3 cr = cross(g_predict_normalize, g_normalize);
        double dt_ = dot(g_predict_normalize, g_normalize);

        M3 R_dg;
        if (std::fabs(1.0 - dt_) < 1e-6) R_dg = m3_identity();
        else {
            M3 sk = skew(cr);
            // I + skew + skew*skew*(1-dot)/(|cross|^2): ((skew*skew)*(1-dot))/den
            double den = cr.x * cr.x + cr.y * cr.y + cr.z * cr.z;
            M3 ss = (sk * sk) * (1.0 - dt_);
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) ss.m[a][b] /= den;
            R_dg = m3_identity() + sk + ss;
        }
        V3 so3_dg = rotation_to_so3(R_dg);
        M32 B_x_predict = derivative_s2(g_predict);
        double d_g[2];
        for (int j = 0; j < 2; j++) d_g[j] = (B_x_predict.m[0][j] * so3_dg.x + B_x_predict.m[1][j] * so3_dg.y) + B_x_predict.m[2][j] * so3_dg.z;

        double d_x[17];
        d_x[0] = d_p.x; d_x[1] = d_p.y; d_x[2] = d_p.z;
        d_x[3] = d_so3.x; d_x[4] = d_so3.y; d_x[5] = d_so3.z;
        d_x[6] = d_v.x; d_x[7] = d_v.y; d_x[8] = d_v.z;
        d_x[9] = d_ba.x; d_x[10] = d_ba.y; d_x[11] = d_ba.z;
        d_x[12] = d_bg.x; d_x[13] = d_bg.y; d_x[14] = d_bg.z;
        d_x[15] = d_g[0]; d_x[16] = d_g[1];

        M3 J_k_so3 = m3_identity() - skew(d_so3) * 0.5;
        // J_k_s2 = I2 + 0.5 * B^T * skew(so3_dg) * B
        auto make_J_s2 = [](const M32 &B, const V3 &w, double J[2][2]) {
            M3 S = skew(w);
            double BtS[2][3];   // (0.5*B^T) * S
            for (int a = 0; a < 2; a++) for (int b = 0; b < 3; b++)
                BtS[a][b] = ((0.5 * B.m[0][a]) * S.m[0][b] + (0.5 * B.m[1][a]) * S.m[1][b]) + (0.5 * B.m[2][a]) * S.m[2][b];
            for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++)
                J[a][b] = ((a == b) ? 1.0 : 0.0) + ((BtS[a][0] * B.m[0][b] + BtS[a][1] * B.m[1][b]) + BtS[a][2] * B.m[2][b]);
        };
        double J_k_s2[2][2];
        make_J_s2(B_x_predict, so3_dg, J_k_s2);

        double d_x_new[17];
        std::memcpy(d_x_new, d_x, sizeof d_x);
        {
            V3 t = J_k_so3 * d_so3;
            d_x_new[3] = t.x; d_x_new[4] = t.y; d_x_new[5] = t.z;
            d_x_new[15] = J_k_s2[0][0] * d_g[0] + J_k_s2[0][1] * d_g[1];
            d_x_new[16] = J_k_s2[1][0] * d_g[0] + J_k_s2[1][1] * d_g[1];
        }

        double covariance[17][17];
        std::memcpy(covariance, eskf_pro->covariance, sizeof covariance);

        auto left3 = [](double C[17][17], const M3 &J, const double S[17][17]) {     // rows 3..5 <- J * rows 3..5 of S
            for (int j = 0; j < 17; j++) {
                V3 c = v3(S[3][j], S[4][j], S[5][j]);
                V3 r = J * c;
                C[3][j] = r.x; C[4][j] = r.y; C[5][j] = r.z;
            }
        };
        auto left2 = [](double C[17][17], const double J[2][2], const double S[17][17]) {
            for (int j = 0; j < 17; j++) {
                double a = S[15][j], b = S[16][j];
                C[15][j] = J[0][0] * a + J[0][1] * b;
                C[16][j] = J[1][0] * a + J[1][1] * b;
            }
        };
        auto right3 = [](double C[17][17], const M3 &J, const double S[17][17]) {    // cols 3..5 <- cols 3..5 of S * J^T
            for (int j = 0; j < 17; j++) {
                double a = S[j][3], b = S[j][4], c = S[j][5];
                for (int k = 0; k < 3; k++) C[j][3 + k] = (a * J.m[k][0] + b * J.m[k][1]) + c * J.m[k][2];
            }
        };
        auto right2 = [](double C[17][17], const double J[2][2], const double S[17][17]) {
            for (int j = 0; j < 17; j++) {
                double a = S[j][15], b = S[j][16];
                C[j][15] = a * J[0][0] + b * J[0][1];
                C[j][16] = a * J[1][0] + b * J[1][1];
            }
        };
        {   // optimize.cpp:222-232 (each product is evaluated into a temporary by Eigen => non-aliased)
            double S[17][17];
            std::memcpy(S, covariance, sizeof S); left3(covariance, J_k_so3, S);
            std::memcpy(S, covariance, sizeof S); left2(covariance, J_k_s2, S);
            std::memcpy(S, covariance, sizeof S); right3(covariance, J_k_so3, S);
            std::memcpy(S, covariance, sizeof S); right2(covariance, J_k_s2, S);
        }

        static double tmpA[289], temp[289], temp_inv[289];
        for (int a = 0; a < 17; a++) for (int b = 0; b < 17; b++) tmpA[a * 17 + b] = covariance[a][b] / laser_point_cov;
        lu_inverse<17>(tmpA, temp);                                        // optimize.cpp:234
        double HTH[36], HTh_seq[6];
        normal_equations(plane_residuals, HTH, HTh_seq);                   // optimize.cpp:235
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) temp[a * 17 + b] += HTH[a * 6 + b];
        lu_inverse<17>(temp, temp_inv);                                    // optimize.cpp:237

        // K_h = temp_inv.block<17,6>(0,0) * H_x^T * h, literally left-to-right: (17x6 * 6xM) * (Mx1)
        double K_h[17];
        for (int a = 0; a < 17; a++) {
            double s = 0.0;
            for (int r = 0; r < M; r++) {
                double e = 0.0;
                for (int c = 0; c < 6; c++) e += temp_inv[a * 17 + c] * H_x[(size_t)r * 6 + c];
                s += e * h[r];
            }
            K_h[a] = s;
        }
        double K_x[17][17];
        std::memset(K_x, 0, sizeof K_x);
        for (int a = 0; a < 17; a++) for (int b = 0; b < 6; b++) {
            double s = 0.0;
            for (int c = 0; c < 6; c++) s += temp_inv[a * 17 + c] * HTH[c * 6 + b];
            K_x[a][b] = s;
        }
        for (int a = 0; a < 17; a++) {                                      // optimize.cpp:244
            double s = 0.0;
            for (int b = 0; b < 17; b++) s += (K_x[a][b] - ((a == b) ? 1.0 : 0.0)) * d_x_new[b];
            d_x[a] = -K_h[a] + s;
        }

        if (log && iters <= max_log_iters) {
            double *L = log + (size_t)(iters - 1) * 61;
            std::memcpy(L, HTH, 36 * sizeof(double));
            std::memcpy(L + 36, HTh_seq, 6 * sizeof(double));
            std::memcpy(L + 42, d_x, 17 * sizeof(double));
            L[59] = (double)M;
            L[60] = loss_old;
        }

        V3 g_before = eskf_pro->g;

        if (norm(v3(d_x[0], d_x[1], d_x[2])) > 100.0 || angular_distance(v3(d_x[3], d_x[4], d_x[5])) > 100.0) {
            continue;                                                       // optimize.cpp:248-251
        }

        eskf_observe(eskf_pro, d_x);                                        // optimize.cpp:253

        f.translation = eskf_pro->p;                                        // optimize.cpp:255-261
        f.rotation = eskf_pro->q;
        velocity = eskf_pro->v;
        ba_out = eskf_pro->ba;
        bg_out = eskf_pro->bg;

        bool converage = false;
        if (f.frame_id > 1 && norm(v3(d_x[0], d_x[1], d_x[2])) < o.threshold_translation_norm &&
            angular_distance(v3(d_x[3], d_x[4], d_x[5])) < o.threshold_orientation_norm) {
            converage = true;
        }

        if (converage || i == max_num_iter - 1) {                           // optimize.cpp:272-310
            double covariance_new[17][17];
            std::memcpy(covariance_new, covariance, sizeof covariance);
            M32 B_x_before = derivative_s2(g_before);
            J_k_so3 = m3_identity() - skew(v3(d_x[3], d_x[4], d_x[5])) * 0.5;
            V3 Bd = v3(B_x_before.m[0][0] * d_x[15] + B_x_before.m[0][1] * d_x[16],
                       B_x_before.m[1][0] * d_x[15] + B_x_before.m[1][1] * d_x[16],
                       B_x_before.m[2][0] * d_x[15] + B_x_before.m[2][1] * d_x[16]);
            make_J_s2(B_x_before, Bd, J_k_s2);

            left3(covariance_new, J_k_so3, covariance);                     // :281-282
            left2(covariance_new, J_k_s2, covariance);                      // :284-285
            {                                                               // :287-291
                double S[17][17];
                std::memcpy(S, covariance, sizeof S);
                right3(covariance_new, J_k_so3, S);
                right3(covariance, J_k_so3, S);
            }
            {                                                               // :293-297
                double S[17][17];
                std::memcpy(S, covariance, sizeof S);
                right2(covariance_new, J_k_s2, S);
                right2(covariance, J_k_s2, S);
            }
            for (int j = 0; j < 6; j++) {                                   // :299-303
                V3 c = v3(K_x[3][j], K_x[4][j], K_x[5][j]);
                V3 r = J_k_so3 * c;
                K_x[3][j] = r.x; K_x[4][j] = r.y; K_x[5][j] = r.z;
            }
            for (int j = 0; j < 6; j++) {
                double a = K_x[15][j], b = K_x[16][j];
                K_x[15][j] = J_k_s2[0][0] * a + J_k_s2[0][1] * b;
                K_x[16][j] = J_k_s2[1][0] * a + J_k_s2[1][1] * b;
            }
            double result[17][17];
            for (int a = 0; a < 17; a++) for (int b = 0; b < 17; b++) {     // :305
                double s = 0.0;
                for (int c = 0; c < 6; c++) s += K_x[a][c] * covariance[c][b];
                result[a][b] = covariance_new[a][b] - s;
            }
            std::memcpy(eskf_pro->covariance, result, sizeof result);       // :307
            break;
        }
    }
    return iters;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

void orc_icp_opts_default(orc_icp_opts *o) {
    o->threshold_voxel_occupancy = 1;
    o->init_num_frames = 20;
    o->size_voxel_map = 1.0;
    o->num_iters_icp = 5;
    o->min_number_neighbors = 20;
    o->voxel_neighborhood = 1;
    o->power_planarity = 2.0;
    o->estimate_normal_from_neighborhood = 1;
    o->max_number_neighbors = 20;
    o->max_dist_to_plane_icp = 0.3;
    o->threshold_orientation_norm = 0.1;
    o->threshold_translation_norm = 0.01;
    o->max_num_residuals = 600;
    o->weight_alpha = 0.9;
    o->weight_neighborhood = 0.1;
}

orc_map *orc_map_create(void) { return new orc_map(); }
void orc_map_destroy(orc_map *m) { delete m; }

// lioOptimization::addPointsToMap (src/lioOptimization.cpp:520-554): rgbPoint(point.point) then addPointToMap
int orc_map_add_points(orc_map *m, const double *xyz, int n, double voxel_size, int max_num_points_in_voxel,
                       double min_distance_points, int min_num_points) {
    int added = 0;
    for (int i = 0; i < n; i++) {
        rgbPoint rgb_point(v3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        if (add_point_to_map(m, rgb_point, voxel_size, max_num_points_in_voxel, min_distance_points, min_num_points)) added++;
    }
    return added;
}
// lioOptimization::mapSize (src/lioOptimization.cpp:574-581)
size_t orc_map_size(const orc_map *m) {
    size_t map_size = 0;
    for (auto it = m->map.begin(); it != m->map.end(); ++it) map_size += ORC_VALUE(it).NumPoints();
    return map_size;
}
int orc_map_num_voxels(const orc_map *m) { return (int)m->map.size(); }
void orc_map_export(const orc_map *m, int cap, int16_t *keys, int32_t *counts, float *xyz) {
    const int V = (int)m->creation_order.size();
    for (int v = 0; v < V; v++) {
        const voxel &k = m->creation_order[v];
        auto it = m->map.find(k);
        const voxelBlock &b = ORC_VALUE(it);
        keys[3 * v] = k.x; keys[3 * v + 1] = k.y; keys[3 * v + 2] = k.z;
        counts[v] = b.NumPoints();
        for (int i = 0; i < cap; i++) {
            float *d = xyz + ((size_t)v * cap + i) * 3;
            if (i < b.NumPoints()) { d[0] = b.points[i].position[0]; d[1] = b.points[i].position[1]; d[2] = b.points[i].position[2]; }
            else { d[0] = d[1] = d[2] = 0.0f; }
        }
    }
}
// ORACLE ADDITION (tests only): rebuild a map from exported arrays, voxel by voxel, WITHOUT the insertion rules of
// addPointToMap -- lets a test hand both sides a map those rules would never produce (e.g. 20 identical points in one
// voxel, whose planarity is NaN: optimize.cpp:348-350).
int orc_map_import(orc_map *m, const int16_t *keys, const int32_t *counts, const float *xyz, int V, int cap) {
    m->map.clear();
    m->creation_order.clear();
    for (int v = 0; v < V; v++) {
        voxel k(keys[3 * v], keys[3 * v + 1], keys[3 * v + 2]);
        if (m->map.find(k) != m->map.end() || counts[v] < 0 || counts[v] > cap) return -1;
        voxelBlock b(cap);
        for (int i = 0; i < counts[v]; i++) {
            const float *p = xyz + ((size_t)v * cap + i) * 3;
            rgbPoint pt(v3(0, 0, 0));                   // position is what the hot path reads (cloudMap.h:54)
            pt.position[0] = p[0]; pt.position[1] = p[1]; pt.position[2] = p[2];
            b.AddPoint(pt);
        }
        b.voxel_index = v;
        m->creation_order.push_back(k);
        m->map[k] = std::move(b);
    }
    return 0;
}
uint64_t orc_voxel_hash(int16_t x, int16_t y, int16_t z) { return (uint64_t)voxel_hash()(voxel(x, y, z)); }
int16_t orc_voxel_coord(double v, double size) { return static_cast<short>(v / size); }
const char *orc_map_backend(void) { return kBackend; }

int orc_search_neighbors(orc_map *m, const double p[3], int nb_voxels_visited, double size_voxel_map,
                         int max_num_neighbors, int threshold_voxel_capacity, int cap, double *out_xyz,
                         int32_t *out_ids, double *out_dist, int *tie_flag, int *num_candidates) {
    NeighborResult r;
    search_neighbors(m->map, v3(p[0], p[1], p[2]), nb_voxels_visited, size_voxel_map, max_num_neighbors,
                     threshold_voxel_capacity, cap, r, true);
    for (size_t i = 0; i < r.pts.size(); i++) {
        if (out_xyz) { out_xyz[3 * i] = r.pts[i].x; out_xyz[3 * i + 1] = r.pts[i].y; out_xyz[3 * i + 2] = r.pts[i].z; }
        if (out_ids) out_ids[i] = r.ids[i];
        if (out_dist) out_dist[i] = r.dist[i];
    }
    if (tie_flag) *tie_flag = r.tie ? 1 : 0;
    if (num_candidates) *num_candidates = r.num_candidates;
    return (int)r.pts.size();
}

int orc_neighborhood(const double *pts, int n, double center[3], double normal[3], double cov[9], double *a2D,
                     double eigenvalues[3]) {
    std::vector<V3> P(n);
    for (int i = 0; i < n; i++) P[i] = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    Neighborhood nb;
    bool ok = compute_neighborhood(P, nb);
    if (center) { center[0] = nb.center.x; center[1] = nb.center.y; center[2] = nb.center.z; }
    if (normal) { normal[0] = nb.normal.x; normal[1] = nb.normal.y; normal[2] = nb.normal.z; }
    if (cov) m3_to(nb.covariance, cov);
    if (a2D) *a2D = nb.a2D;
    if (eigenvalues) { eigenvalues[0] = nb.evals[0]; eigenvalues[1] = nb.evals[1]; eigenvalues[2] = nb.evals[2]; }
    return ok ? 0 : -1;
}

int orc_build_plane_residuals(orc_map *m, const orc_icp_opts *o, const double *raw_xyz, int n,
                              const double q_wxyz[4], const double t[3], const double t_last[3],
                              const double R_il[9], const double t_il[3], int frame_id, int cap,
                              orc_residual_out *out, orc_normal_eq *neq) {
    FrameCtx f;
    f.rotation = Q4{q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]};
    f.translation = v3(t[0], t[1], t[2]);
    f.last_translation = v3(t_last[0], t_last[1], t_last[2]);
    f.R_imu_lidar = m3_from(R_il);
    f.t_imu_lidar = v3(t_il[0], t_il[1], t_il[2]);
    f.frame_id = frame_id;
    std::vector<planeParam> pr;
    double loss = 0.0;
    orc_normal_eq local;
    if (!neq) neq = &local;
    std::memset(neq, 0, sizeof *neq);
    int rc = build_plane_residuals(m, *o, raw_xyz, n, f, cap, pr, loss, out, neq);
    if (rc != 0) return rc;
    normal_equations(pr, neq->HtH, neq->Hth);
    return 0;
}

orc_eskf *orc_eskf_create(void) { orc_eskf *e = new orc_eskf; eskf_init(e); return e; }
void orc_eskf_destroy(orc_eskf *e) { delete e; }
void orc_eskf_get_state(const orc_eskf *e, double s[19]) {
    s[0] = e->p.x; s[1] = e->p.y; s[2] = e->p.z;
    s[3] = e->q.w; s[4] = e->q.x; s[5] = e->q.y; s[6] = e->q.z;
    s[7] = e->v.x; s[8] = e->v.y; s[9] = e->v.z;
    s[10] = e->ba.x; s[11] = e->ba.y; s[12] = e->ba.z;
    s[13] = e->bg.x; s[14] = e->bg.y; s[15] = e->bg.z;
    s[16] = e->g.x; s[17] = e->g.y; s[18] = e->g.z;
}
void orc_eskf_set_state(orc_eskf *e, const double s[19]) {
    e->p = v3(s[0], s[1], s[2]);
    e->q = Q4{s[3], s[4], s[5], s[6]};
    e->v = v3(s[7], s[8], s[9]);
    e->ba = v3(s[10], s[11], s[12]);
    e->bg = v3(s[13], s[14], s[15]);
    e->g = v3(s[16], s[17], s[18]);
}
void orc_eskf_get_cov(const orc_eskf *e, double P[289]) { std::memcpy(P, e->covariance, 289 * sizeof(double)); }
void orc_eskf_set_cov(orc_eskf *e, const double P[289]) { std::memcpy(e->covariance, P, 289 * sizeof(double)); }
// setAccCov.. + initializeNoise (eskfEstimator.cpp:23-41,120-126): after tryInit gyr_cov=gyr_cov_scale etc.
int orc_eskf_try_init(orc_eskf *e, const double *t, const double *gyr, const double *acc, int n) {     // eskfEstimator.cpp:43-91
    if (n <= 0) return e->initial_flag ? 1 : 0;     // imu_meas.front()/back() on an empty vector is UB upstream
    eskf_initialization(e, t, gyr, acc, n);
    if (e->num_init_meas > 10 /*MIN_INI_COUNT*/ && t[n - 1] - e->time_first_imu > 3.0 /*MIN_INI_TIME*/) {
        const double r = e->G_norm / norm(e->mean_acc);
        e->acc_cov = e->acc_cov * std::pow(r, 2);
        if (norm(e->gyr_cov) > 0.5 /*MAX_GYR_VAR*/) return -1;
        if (norm(e->acc_cov) > 0.6 /*MAX_ACC_VAR*/) return -2;
        e->initial_flag = true;
        e->gyr_cov = e->gyr_cov_scale;
        e->acc_cov = e->acc_cov_scale;
        e->bg = e->mean_gyr;
        e->g = (e->mean_acc / norm(e->mean_acc)) * e->G_norm;
        orc_eskf_scale_init_cov(e);
        std::memset(e->noise, 0, sizeof e->noise);                                                    // initializeNoise :120-126
        for (int i = 0; i < 3; i++) {
            const double a[3] = {e->acc_cov.x, e->acc_cov.y, e->acc_cov.z}, g[3] = {e->gyr_cov.x, e->gyr_cov.y, e->gyr_cov.z};
            const double ba[3] = {e->b_acc_cov.x, e->b_acc_cov.y, e->b_acc_cov.z}, bg[3] = {e->b_gyr_cov.x, e->b_gyr_cov.y, e->b_gyr_cov.z};
            e->noise[i][i] = a[i]; e->noise[3 + i][3 + i] = g[i]; e->noise[6 + i][6 + i] = ba[i]; e->noise[9 + i][9 + i] = bg[i];
        }
        return 1;
    }
    return 0;
}
void orc_eskf_get_init_stats(const orc_eskf *e, double out[14]) {
    const V3 *v[4] = {&e->mean_gyr, &e->mean_acc, &e->gyr_cov, &e->acc_cov};
    for (int k = 0; k < 4; k++) { out[3 * k] = v[k]->x; out[3 * k + 1] = v[k]->y; out[3 * k + 2] = v[k]->z; }
    out[12] = (double)e->num_init_meas;
    out[13] = e->initial_flag ? 1.0 : 0.0;
}
void orc_eskf_set_g_norm(orc_eskf *e, double g_norm) { e->G_norm = g_norm; }

// stateInitialization (lioOptimization.cpp:895-990).  prev2/prev1: (q wxyz, t) of all_cloud_frame[size-2] / [size-1].
// initialization (utility.h:88-92): 0 INIT_IMU, 1 INIT_CONSTANT_VELOCITY, anything else -> copy the last pose
void orc_state_initialization(int index_frame, int initialization, int initial_flag, const double prev2[7], const double prev1[7],
                              const double eskf_q[4], const double eskf_t[3], double out[7]) {
    Q4 q = Q4{1, 0, 0, 0}; V3 t = v3(0, 0, 0);
    if (index_frame > 2) {
        const Q4 q1 = Q4{prev1[0], prev1[1], prev1[2], prev1[3]}, q2 = Q4{prev2[0], prev2[1], prev2[2], prev2[3]};
        const V3 t1 = v3(prev1[4], prev1[5], prev1[6]), t2 = v3(prev2[4], prev2[5], prev2[6]);
        const bool const_vel = initialization == 1 || (initialization == 0 && !initial_flag);
        if (const_vel) {
            const Q4 d = q_mul(q1, q_inverse(q2));                // (q1 * q2^-1) * q1: operator* is left-associative
            q = q_mul(d, q1);
            t = t1 + q_rotate(d, t1 - t2);                        // Quaternion * Vector3 = rotate by the (q1 q2^-1) product
        } else if (initialization == 0) {
            q = Q4{eskf_q[0], eskf_q[1], eskf_q[2], eskf_q[3]};
            t = v3(eskf_t[0], eskf_t[1], eskf_t[2]);
        } else {
            q = q1; t = t1;
        }
    }
    out[0] = q.w; out[1] = q.x; out[2] = q.y; out[3] = q.z; out[4] = t.x; out[5] = t.y; out[6] = t.z;
}

void orc_eskf_set_noise(orc_eskf *e, double acc_cov, double gyr_cov, double b_acc_cov, double b_gyr_cov) {
    e->acc_cov_scale = v3(acc_cov, acc_cov, acc_cov);
    e->gyr_cov_scale = v3(gyr_cov, gyr_cov, gyr_cov);
    e->acc_cov = v3(acc_cov, acc_cov, acc_cov);
    e->gyr_cov = v3(gyr_cov, gyr_cov, gyr_cov);
    e->b_acc_cov = v3(b_acc_cov, b_acc_cov, b_acc_cov);
    e->b_gyr_cov = v3(b_gyr_cov, b_gyr_cov, b_gyr_cov);
    std::memset(e->noise, 0, sizeof e->noise);
    const double d[12] = {acc_cov, acc_cov, acc_cov, gyr_cov, gyr_cov, gyr_cov,
                          b_acc_cov, b_acc_cov, b_acc_cov, b_gyr_cov, b_gyr_cov, b_gyr_cov};
    for (int i = 0; i < 12; i++) e->noise[i][i] = d[i];
}
void orc_eskf_init_imu(orc_eskf *e, const double acc0[3], const double gyr0[3]) {
    e->acc_0 = v3(acc0[0], acc0[1], acc0[2]);
    e->gyr_0 = v3(gyr0[0], gyr0[1], gyr0[2]);
}
void orc_eskf_scale_init_cov(orc_eskf *e) {       // eskfEstimator.cpp:74-76
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        e->covariance[9 + i][9 + j] *= 0.001;
        e->covariance[12 + i][12 + j] *= 0.0001;
    }
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) e->covariance[15 + i][15 + j] *= 0.00001;
}
void orc_eskf_predict(orc_eskf *e, double dt, const double acc1[3], const double gyr1[3]) {
    eskf_predict(e, dt, v3(acc1[0], acc1[1], acc1[2]), v3(gyr1[0], gyr1[1], gyr1[2]));
}
void orc_eskf_observe(orc_eskf *e, const double dx[17]) { eskf_observe(e, dx); }

int orc_update_iekf(orc_map *m, orc_eskf *e, const orc_icp_opts *o, const double *raw_xyz, int n,
                    double state_io[16], const double t_last[3], const double R_il[9], const double t_il[3],
                    int frame_id, int cap, double laser_point_cov, double *log, int max_log_iters,
                    int *num_residuals_used) {
    FrameCtx f;
    f.rotation = Q4{state_io[0], state_io[1], state_io[2], state_io[3]};
    f.translation = v3(state_io[4], state_io[5], state_io[6]);
    V3 vel = v3(state_io[7], state_io[8], state_io[9]);
    V3 ba = v3(state_io[10], state_io[11], state_io[12]);
    V3 bg = v3(state_io[13], state_io[14], state_io[15]);
    f.last_translation = v3(t_last[0], t_last[1], t_last[2]);
    f.R_imu_lidar = m3_from(R_il);
    f.t_imu_lidar = v3(t_il[0], t_il[1], t_il[2]);
    f.frame_id = frame_id;
    int rc = update_iekf(m, e, *o, raw_xyz, n, f, vel, ba, bg, cap, laser_point_cov, log, max_log_iters,
                         num_residuals_used);
    state_io[0] = f.rotation.w; state_io[1] = f.rotation.x; state_io[2] = f.rotation.y; state_io[3] = f.rotation.z;
    state_io[4] = f.translation.x; state_io[5] = f.translation.y; state_io[6] = f.translation.z;
    state_io[7] = vel.x; state_io[8] = vel.y; state_io[9] = vel.z;
    state_io[10] = ba.x; state_io[11] = ba.y; state_io[12] = ba.z;
    state_io[13] = bg.x; state_io[14] = bg.y; state_io[15] = bg.z;
    return rc;
}

void orc_quat_to_rot(const double q[4], double R[9]) { m3_to(quat_to_rot(Q4{q[0], q[1], q[2], q[3]}), R); }
void orc_rot_to_quat(const double R[9], double q[4]) { Q4 r = rot_to_quat(m3_from(R)); q[0] = r.w; q[1] = r.x; q[2] = r.y; q[3] = r.z; }
void orc_transform_points(const double *raw_xyz, int n, const double q[4], const double t[3], const double R_il[9],
                          const double t_il[3], double *world_xyz) {
    const M3 R = quat_to_rot(Q4{q[0], q[1], q[2], q[3]});       // utility.cpp:317: q_end.toRotationMatrix()
    const M3 Ril = m3_from(R_il);
    const V3 til = v3(t_il[0], t_il[1], t_il[2]), te = v3(t[0], t[1], t[2]);
    for (int i = 0; i < n; i++) {
        const V3 raw = v3(raw_xyz[3 * (size_t)i], raw_xyz[3 * (size_t)i + 1], raw_xyz[3 * (size_t)i + 2]);
        const V3 w = R * (Ril * raw + til) + te;
        world_xyz[3 * (size_t)i] = w.x; world_xyz[3 * (size_t)i + 1] = w.y; world_xyz[3 * (size_t)i + 2] = w.z;
    }
}

int orc_grid_sampling(const double *world_xyz, int n, double size_voxel, int32_t *idx_out) {
    // subSampleFrame (utility.cpp:167-186): every point is pushed into its voxel's vector, then the first
    // element of each vector is taken while iterating the tr1 container.
    std::tr1::unordered_map<voxel, std::vector<int>, voxel_hash> grid;
    for (int i = 0; i < n; i++) {
        const short kx = static_cast<short>(world_xyz[3 * (size_t)i] / size_voxel);
        const short ky = static_cast<short>(world_xyz[3 * (size_t)i + 1] / size_voxel);
        const short kz = static_cast<short>(world_xyz[3 * (size_t)i + 2] / size_voxel);
        grid[voxel(kx, ky, kz)].push_back(i);
    }
    int m = 0;
    for (const auto &kv : grid)
        if (kv.second.size() > 0) idx_out[m++] = kv.second[0];
    return m;
}

namespace {
struct ImuState { double timestamp; V3 un_acc, un_gyr, trans; Q4 quat; V3 vel; };
inline ImuState imu_state_at(const double *s, int i) {
    const double *p = s + 17 * (size_t)i;
    ImuState r;
    r.timestamp = p[0];
    r.un_acc = v3(p[1], p[2], p[3]); r.un_gyr = v3(p[4], p[5], p[6]); r.trans = v3(p[7], p[8], p[9]);
    r.quat = Q4{p[10], p[11], p[12], p[13]}; r.vel = v3(p[14], p[15], p[16]);
    return r;
}
// Eigen::QuaternionBase::slerp (Eigen 3.3 Geometry/Quaternion.h)
inline Q4 q_slerp(const Q4 &a, double t, const Q4 &b) {
    const double one = 1.0 - 2.220446049250313e-16;
    const double d = ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w;
    const double absD = std::fabs(d);
    double scale0, scale1;
    if (absD >= one) { scale0 = 1.0 - t; scale1 = t; }
    else {
        const double theta = std::acos(absD), sinTheta = std::sin(theta);
        scale0 = std::sin((1.0 - t) * theta) / sinTheta;
        scale1 = std::sin(t * theta) / sinTheta;
    }
    if (d < 0.0) scale1 = -scale1;
    return Q4{scale0 * a.w + scale1 * b.w, scale0 * a.x + scale1 * b.x, scale0 * a.y + scale1 * b.y, scale0 * a.z + scale1 * b.z};
}
inline V3 p3(const double *a, int i) { return v3(a[3 * (size_t)i], a[3 * (size_t)i + 1], a[3 * (size_t)i + 2]); }
inline void s3(double *a, int i, const V3 &v) { a[3 * (size_t)i] = v.x; a[3 * (size_t)i + 1] = v.y; a[3 * (size_t)i + 2] = v.z; }
}  // namespace

void orc_distort_frame_by_constant(const double *raw_xyz, const double *relative_time, int n, const double *imu_states,
                                   int n_states, double time_frame_begin, const double R_il[9], const double t_il[3],
                                   double *imu_point) {                                     // utility.cpp:203-236
    const ImuState first = imu_state_at(imu_states, 0), last = imu_state_at(imu_states, n_states - 1);
    const double time_frame_end = last.timestamp;
    const M3 Ril = m3_from(R_il);
    const V3 til = v3(t_il[0], t_il[1], t_il[2]);
    for (int i = 0; i < n; i++) {
        double time_point = time_frame_begin + relative_time[i] / 1000.0;
        if (std::fabs(time_point - time_frame_begin) < 1e-6) time_point = time_frame_begin + 1e-6;
        if (std::fabs(time_point - time_frame_end) < 1e-6) time_point = time_frame_end - 1e-6;
        double alpha_time = (time_point - time_frame_begin) / (time_frame_end - time_frame_begin);
        if (alpha_time > 1) alpha_time = 1;
        if (alpha_time < 0) alpha_time = 0;
        const Q4 quat_alpha = q_slerp(first.quat, alpha_time, last.quat);
        const V3 trans_alpha = (1.0 - alpha_time) * first.trans + alpha_time * last.trans;
        s3(imu_point, i, quat_to_rot(quat_alpha) * (Ril * p3(raw_xyz, i) + til) + trans_alpha);
    }
}

int orc_distort_frame_by_imu(const double *raw_xyz, const double *relative_time, int n, const double *imu_states,
                             int n_states, double time_frame_begin, const double R_il[9], const double t_il[3],
                             double *imu_point) {                                           // utility.cpp:238-306
    const M3 Ril = m3_from(R_il);
    const V3 til = v3(t_il[0], t_il[1], t_il[2]);
    int iter = 0;
    for (int k = 0; k + 1 < n_states; k++) {
        const ImuState a = imu_state_at(imu_states, k), b = imu_state_at(imu_states, k + 1);
        const double time_imu_begin = a.timestamp, time_imu_end = b.timestamp;
        while (iter != n) {
            double time_point = time_frame_begin + relative_time[iter] / 1000.0;
            if (time_point > time_imu_begin - 1e-6 && time_point < time_imu_end + 1e-6) {
                if (std::fabs(time_point - time_imu_begin) < 1e-6) time_point = time_imu_begin + 1e-6;
                if (std::fabs(time_point - time_imu_end) < 1e-6) time_point = time_imu_end - 1e-6;
                const double dt = time_point - time_imu_begin;
                const Q4 quat_point = q_normalized(q_mul(a.quat, so3_to_quat(b.un_gyr * dt)));
                const V3 trans_point = (a.trans + a.vel * dt) + ((0.5 * b.un_acc) * dt) * dt;
                s3(imu_point, iter, quat_to_rot(quat_point) * (Ril * p3(raw_xyz, iter) + til) + trans_point);
                iter++;
            } else {
                break;
            }
        }
    }
    return iter;
}

void orc_transform_all_imu_point(const double *imu_point, int n, const double *imu_states, int n_states,
                                 const double R_il[9], const double t_il[3], double *raw_xyz) {   // utility.cpp:320-332
    const ImuState last = imu_state_at(imu_states, n_states - 1);
    const M3 Rinv = quat_to_rot(q_inverse(last.quat));
    const V3 trans_end_inv = (Rinv * -1.0) * last.trans;
    const M3 RilT = transpose(m3_from(R_il));
    const V3 til = v3(t_il[0], t_il[1], t_il[2]);
    for (int i = 0; i < n; i++) s3(raw_xyz, i, RilT * (Rinv * p3(imu_point, i) + trans_end_inv) - RilT * til);
}

int orc_build_frame_order(const double *point_xyz, int n, double sample_size, int do_subsample, int32_t *order_out) {
    std::vector<int> frame((size_t)n);
    for (int i = 0; i < n; i++) frame[i] = i;
    std::mt19937_64 seed;                                          // boost::mt19937_64 seed; (lioOptimization.cpp:840)
    std::shuffle(frame.begin(), frame.end(), seed);
    if (do_subsample) {                                            // odometry_options.voxel_size > 0
        std::tr1::unordered_map<voxel, std::vector<int>, voxel_hash> grid;      // subSampleFrame, utility.cpp:167-186
        for (int i = 0; i < (int)frame.size(); i++) {
            const double *p = point_xyz + 3 * (size_t)frame[i];
            grid[voxel(static_cast<short>(p[0] / sample_size), static_cast<short>(p[1] / sample_size),
                       static_cast<short>(p[2] / sample_size))].push_back(frame[i]);
        }
        frame.resize(0);
        for (const auto &kv : grid)
            if (kv.second.size() > 0) frame.push_back(kv.second[0]);
        std::shuffle(frame.begin(), frame.end(), seed);
    }
    for (size_t i = 0; i < frame.size(); i++) order_out[i] = frame[i];
    return (int)frame.size();
}

int orc_make_point_timestamp(const double *timestamp, int n, double time_begin, double time_end, int point_time_enable,
                             double *relative_time, double *alpha_time, uint8_t *keep_out) {      // lioOptimization.cpp:786-819
    const double delta_t = time_end - time_begin;
    int kept = 0;
    for (int i = 0; i < n; i++) {
        if (!point_time_enable && (timestamp[i] > time_end || timestamp[i] < time_begin)) { keep_out[i] = 0; continue; }
        keep_out[i] = 1; kept++;
        relative_time[i] = timestamp[i] - time_begin;
        alpha_time[i] = relative_time[i] / delta_t;
        relative_time[i] = relative_time[i] * 1000.0;
        if (point_time_enable && alpha_time[i] > 1.0) alpha_time[i] = 1.0 - 1e-5;
    }
    return kept;
}

uint64_t orc_mt19937_64_nth(int nth) {
    std::mt19937_64 e;
    uint64_t v = 0;
    for (int i = 0; i < nth; i++) v = e();
    return v;
}

void orc_so3_to_rot(const double w[3], double R[9]) { m3_to(so3_to_rotation(v3(w[0], w[1], w[2])), R); }
void orc_so3_to_quat(const double w[3], double q[4]) { Q4 r = so3_to_quat(v3(w[0], w[1], w[2])); q[0] = r.w; q[1] = r.x; q[2] = r.y; q[3] = r.z; }
void orc_rot_to_so3(const double R[9], double w[3]) { V3 r = rotation_to_so3(m3_from(R)); w[0] = r.x; w[1] = r.y; w[2] = r.z; }
double orc_angular_distance_so3(const double w[3]) { return angular_distance(v3(w[0], w[1], w[2])); }
void orc_derivative_s2(const double g[3], double B[6]) { M32 b = derivative_s2(v3(g[0], g[1], g[2])); for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) B[2 * i + j] = b.m[i][j]; }
int orc_inverse17(const double A[289], double Ainv[289]) { return lu_inverse<17>(A, Ainv) ? 0 : -1; }
void orc_eig3(const double A[9], double evals[3], double evecs[9]) {
    double a[3][3], V[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = A[3 * i + j];
    if (g_eig_solver == 1) eig3_jacobi(a, evals, V); else eig3_eigen_ql(a, evals, V);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) evecs[3 * i + j] = V[i][j];
}
int orc_eig3_solver(int solver, const double A[9], double evals[3], double evecs[9]) {
    double a[3][3], V[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = A[3 * i + j];
    bool ok = true;
    if (solver == 1) eig3_jacobi(a, evals, V); else ok = eig3_eigen_ql(a, evals, V);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) evecs[3 * i + j] = V[i][j];
    return ok ? 0 : -1;
}
void orc_set_eig_solver(int solver) { g_eig_solver = solver == 1 ? 1 : 0; }
void orc_set_threads(int threads) { g_threads = threads < 1 ? 1 : threads; }
int  orc_get_threads(void) { return g_threads; }
int  orc_get_eig_solver(void) { return g_eig_solver; }

// the literal bounded priority queue of searchNeighbors (optimize.cpp:355-363, 394-404, 411-422) over a plain list of
// distances offered in index order: read-out order of the surviving indices (ascending distance).  Returns their number.
int orc_heap_topk(const double *distances, int n, int max_num_neighbors, int32_t *out_index) {
    priority_queue_t priority_queue;
    const voxel vx(0, 0, 0);
    for (int i = 0; i < n; i++) {
        double distance = distances[i];
        if ((int)priority_queue.size() == max_num_neighbors) {
            if (distance < std::get<0>(priority_queue.top())) {
                priority_queue.pop();
                priority_queue.emplace(distance, v3(0, 0, 0), vx, i);
            }
        } else {
            priority_queue.emplace(distance, v3(0, 0, 0), vx, i);
        }
    }
    auto size = priority_queue.size();
    for (size_t i = 0; i < size; ++i) {
        out_index[size - 1 - i] = std::get<3>(priority_queue.top());
        priority_queue.pop();
    }
    return (int)size;
}

}  // extern "C"
