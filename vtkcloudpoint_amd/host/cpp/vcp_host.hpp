// vcp_host.hpp -- C++ host-side mirror of the reference's C# class surface over the C-ABI (include/vcp.h).
//
// The reference is compiled code (C# / .NET 3.5) and no C# toolchain exists in this image, so the host side
// above the C-ABI is C++: same class and member names, same argument meaning, same in-place mutation of the
// caller's Point3D objects, errors as exceptions (the C# throws too).  The C# sources a maintainer would
// drop into vtkPointCloud/BaseClass/ are in ../csharp/ and INTEGRATION.md.
//   Point3D    BaseClass/DataModel.cs:102-160        DBImproved  BaseClass/DBImproved.cs:8-116
//   Matrix     BaseClass/Matrix.cs (slice ICP uses)  ICP         BaseClass/ICP.cs:8-314
//   Tools      BaseClass/Tools.cs:162-195, :580-621  MainFormPath FrmMain.cs:1214-1291,:1432-1544,:3572-3618
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "vcp.h"

namespace vtkPointCloud {

struct VcpException : std::runtime_error {
  int code;
  VcpException(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

class Context {
 public:
  explicit Context(int device = 0) {
    int rc = vcp_create(device, &ctx_);
    if (rc != VCP_OK) throw VcpException(rc, vcp_last_error(nullptr));
  }
  ~Context() { vcp_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  vcp_ctx* get() const { return ctx_; }
  void check(int rc) const {
    if (rc != VCP_OK) throw VcpException(rc, vcp_last_error(ctx_));
  }
  // give the device workspace back (it is re-allocated by the next call)
  void release_workspace() { check(vcp_release_workspace(ctx_)); }

 private:
  vcp_ctx* ctx_ = nullptr;
};

struct Point3D {  // DataModel.cs:120-144
  int IDBeforeMerge = 0;
  double motor_x = 0, motor_y = 0, Distance = 0, X = 0, Y = 0, Z = 0;
  int clusterId = 0, pathId = 0;
  bool ifShown = false;
  int ptsCount = 0;
  bool isClassed = false, isKeyPoint = false, isMatched = false;
  int matchNum = 0;
  double tmp_X = 0, tmp_Y = 0, tmp_Z = 0, matched_X = 0, matched_Y = 0, matched_Z = 0;
  Point3D() = default;
  Point3D(double xx, double yy, double zz, int id = 0, bool shown = false) : X(xx), Y(yy), Z(zz), clusterId(id), ifShown(shown) {}
};

struct ClusObj {  // DataModel.cs:14-33
  std::vector<Point3D*> li;
  int clusId = 0;
  bool visible = true;
};

class DBImproved {
 public:
  int clusterAmount = 0;                   // DBImproved.cs:10
  int pointsAmount = 0;                    // :11
  static inline long long iritatorNum = 0; // :12 (64-bit here; the C# int overflows past 2.1e9)
  int cf = 0;                              // :13
  explicit DBImproved(Context& c) : c_(c) {}

  static double getDisP(const Point3D& p1, const Point3D& p2) {  // :14-25
    double dx = p1.motor_x - p2.motor_x, dy = p1.motor_y - p2.motor_y;
    iritatorNum++;
    return std::fabs(dx) + std::fabs(dy);
  }

  void dbscan(std::vector<Point3D*>& lst, double e, int minPts) {  // :91-114
    const int64_t n = (int64_t)lst.size();
    if (n == 0) {
      clusterAmount = cf;
      return;
    }
    std::vector<double> xy(2 * n);
    std::vector<uint8_t> cls(n), core(n), out_cls(n);
    std::vector<int32_t> lab(n);
    bool any = false;
    for (int64_t i = 0; i < n; i++) {
      xy[2 * i] = lst[i]->motor_x;
      xy[2 * i + 1] = lst[i]->motor_y;
      cls[i] = lst[i]->isClassed;
      lab[i] = lst[i]->clusterId;
      any |= lst[i]->isClassed;
    }
    int32_t cf_out = 0;
    int64_t ev = 0;
    c_.check(vcp_dbscan(c_.get(), xy.data(), n, 2, VCP_L1_2D, e, minPts, cf, nullptr, any ? cls.data() : nullptr,
                        lab.data(), core.data(), out_cls.data(), &cf_out, &ev));
    for (int64_t i = 0; i < n; i++) {
      if (any || lab[i] != 0) lst[i]->clusterId = lab[i];
      if (out_cls[i]) lst[i]->isClassed = true;
      if (core[i]) lst[i]->isKeyPoint = true;
    }
    pointsAmount += (int)n;
    cf = cf_out;
    clusterAmount = cf;
    iritatorNum += ev;
  }

 private:
  Context& c_;
};

class Matrix {  // BaseClass/Matrix.cs:18-34
 public:
  int rows, cols;
  std::vector<double> mat;
  Matrix(int r, int c) : rows(r), cols(c), mat((size_t)r * c, 0.0) {}
  double& operator()(int r, int c) { return mat.at((size_t)r * cols + c); }
  double operator()(int r, int c) const { return mat.at((size_t)r * cols + c); }
};

class ICP {  // BaseClass/ICP.cs:8
 public:
  explicit ICP(Context& c) : c_(c) {}
  int max_iter = 1000;
  // ICP.cs:18: model = truth, data = measurements; R (3x3) and T (3x1) are written in place
  void go_hell_ICP(const std::vector<Point3D*>& model, const std::vector<Point3D*>& data, Matrix& R, Matrix& T, double e) {
    if (R.rows != 3 || R.cols != 3 || T.rows != 3 || T.cols != 1) throw VcpException(VCP_ERR_ARG, "R 3x3, T 3x1");
    if (data.empty()) return;
    std::vector<double> m(3 * model.size()), d(3 * data.size());
    for (size_t i = 0; i < model.size(); i++) { m[3 * i] = model[i]->X; m[3 * i + 1] = model[i]->Y; m[3 * i + 2] = model[i]->Z; }
    for (size_t i = 0; i < data.size(); i++) { d[3 * i] = data[i]->X; d[3 * i + 1] = data[i]->Y; d[3 * i + 2] = data[i]->Z; }
    double r[9], t[3], sse = 0, rmse = 0;
    int32_t it = 0;
    c_.check(vcp_icp(c_.get(), m.data(), (int64_t)model.size(), d.data(), (int64_t)data.size(), e, max_iter,
                     VCP_STOP_SSE_DELTA, r, t, &sse, &rmse, &it));
    last_sse = sse;
    last_iters = it;
    if (it == 1 && sse < e) return;  // the C# never touches R,T when round 1 already meets the stop rule
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R(i, j) = r[3 * i + j];
      T(i, 0) = t[i];
    }
  }
  double last_sse = 0;
  int last_iters = 0;

 private:
  Context& c_;
};

struct Tools {
  // Tools.cs:162-195
  static void GetClusList(Context& c, const std::vector<Point3D*>& rawData, std::vector<Point3D>& centers,
                          std::vector<Point3D>& centers2D, std::vector<ClusObj>& clusList) {
    const int64_t n = (int64_t)rawData.size();
    const int32_t K = (int32_t)clusList.size();
    std::vector<double> xyz(3 * n), mot(2 * n);
    std::vector<int32_t> lab(n);
    for (int64_t i = 0; i < n; i++) {
      Point3D* p = rawData[i];
      xyz[3 * i] = p->X; xyz[3 * i + 1] = p->Y; xyz[3 * i + 2] = p->Z;
      mot[2 * i] = p->motor_x; mot[2 * i + 1] = p->motor_y;
      lab[i] = p->clusterId;
      if (p->clusterId != 0) clusList.at(p->clusterId - 1).li.push_back(p);
    }
    if (K == 0 || n == 0) return;
    std::vector<double> c3(3 * K), c2(2 * K);
    std::vector<int64_t> cnt(K);
    c.check(vcp_centroids(c.get(), xyz.data(), mot.data(), lab.data(), n, K, c3.data(), c2.data(), cnt.data()));
    for (int32_t k = 0; k < K; k++) {
      if (cnt[k] == 0) continue;  // :191
      centers.emplace_back(c3[3 * k], c3[3 * k + 1], c3[3 * k + 2], clusList[k].clusId, true);
      centers2D.emplace_back(c2[2 * k], c2[2 * k + 1], 0.0, clusList[k].clusId, true);
    }
  }
  // Tools.cs:580-621
  static std::map<int, int> MergeIDByDistance(Context& c, std::vector<Point3D>& centers, double thre) {
    std::map<int, int> dick;
    const int32_t K = (int32_t)centers.size();
    if (K == 0) return dick;
    std::vector<double> cxy(2 * K);
    std::vector<int32_t> ids(K), map_to(K);
    for (int32_t k = 0; k < K; k++) {
      Point3D& p = centers[k];
      p.IDBeforeMerge = p.clusterId;
      p.motor_x = p.X; p.motor_y = p.Y;
      p.clusterId = 0;
      cxy[2 * k] = p.X; cxy[2 * k + 1] = p.Y;
      ids[k] = p.IDBeforeMerge;
    }
    int32_t mc = 0;
    c.check(vcp_merge_centroids(c.get(), cxy.data(), ids.data(), K, thre, map_to.data(), &mc));
    for (int32_t k = 0; k < K; k++)
      if (map_to[k] != 0) dick[ids[k]] = map_to[k];
    return dick;
  }
};

// MainForm.getClusterFromMotor + DoWork3/StartCode + CompleteWork3 (FrmMain.cs:1214-1291, :2782-2794, :1432-1520)
struct BlockResult {
  std::vector<int64_t> clusForMerge;  // original indices in final order
  int rows = 0, cols = 0, kept = 0, delSum = 0, clusterAmount = 0;
  long long distEvals = 0;
};
inline BlockResult getClusterFromMotor(Context& c, std::vector<Point3D*>& rawData, double tr, int pts, int ptsInCell) {
  const int64_t n = (int64_t)rawData.size();
  std::vector<double> mot(2 * n);
  for (int64_t i = 0; i < n; i++) { mot[2 * i] = rawData[i]->motor_x; mot[2 * i + 1] = rawData[i]->motor_y; }
  std::vector<int32_t> lab(n), blk(n);
  BlockResult r;
  r.clusForMerge.resize(n > 0 ? n : 1);
  int64_t m = 0, ev = 0;
  int32_t rows, cols, kept, del, ca;
  c.check(vcp_dbscan_blocks(c.get(), mot.data(), n, tr, pts, ptsInCell, 3, lab.data(), blk.data(), r.clusForMerge.data(),
                            &m, &rows, &cols, &kept, &del, &ca, &ev));
  r.clusForMerge.resize(m);
  r.rows = rows; r.cols = cols; r.kept = kept; r.delSum = del; r.clusterAmount = ca; r.distEvals = ev;
  for (int64_t i = 0; i < n; i++) {
    rawData[i]->clusterId = lab[i];
    rawData[i]->isClassed = lab[i] != 0;
  }
  return r;
}

// calMatchedCoords + RecorrectMatchingPtsByDistance (FrmMain.cs:3572-3618)
inline int RecorrectMatchingPtsByDistance(Context& c, std::vector<Point3D>& centers, const std::vector<double>& truths,
                                          const double M[16], double matchDistance) {
  const int32_t K = (int32_t)centers.size(), T = (int32_t)(truths.size() / 3);
  std::vector<double> cen(3 * K), mx(3 * K), nd(K);
  std::vector<uint8_t> ok(K);
  std::vector<int32_t> nn(K);
  for (int32_t j = 0; j < K; j++) { cen[3 * j] = centers[j].tmp_X; cen[3 * j + 1] = centers[j].tmp_Y; cen[3 * j + 2] = centers[j].tmp_Z; }
  int32_t cnt = 0;
  c.check(vcp_match(c.get(), cen.data(), K, truths.data(), T, M, matchDistance, mx.data(), ok.data(), nn.data(), nd.data(), &cnt));
  for (int32_t j = 0; j < K; j++) {
    centers[j].matched_X = mx[3 * j]; centers[j].matched_Y = mx[3 * j + 1]; centers[j].matched_Z = mx[3 * j + 2];
    centers[j].isMatched = ok[j];
    if (ok[j]) centers[j].matchNum = nn[j];
  }
  return cnt;
}

}  // namespace vtkPointCloud
