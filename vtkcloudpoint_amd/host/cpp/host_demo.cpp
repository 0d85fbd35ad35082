// host_demo.cpp -- drives the C++ mirror the way FrmMain drives the C# classes; checks the hand-derived
// known answers of tests/golden/micro_cases.json (border point -> higher id; cf preset) and an ICP recovery.
// Build: g++ -std=c++17 -I include host_demo.cpp -L vtkcloudpoint_amd -lvcp -Wl,-rpath,<dir>
#include <cstdio>
#include <cmath>
#include <memory>

#include "vcp_host.hpp"

using namespace vtkPointCloud;

int main() {
  try {
    Context ctx(0);
    const double xs[9] = {0.0, 0.25, 0.5, 0.75, 1.75, 2.75, 3.0, 3.25, 3.5};
    std::vector<std::unique_ptr<Point3D>> own;
    std::vector<Point3D*> lst;
    for (double x : xs) {
      own.emplace_back(new Point3D());
      own.back()->motor_x = x;
      lst.push_back(own.back().get());
    }
    DBImproved db(ctx);
    db.cf = 5;  // FrmMain.cs:1509
    db.dbscan(lst, 1.0, 4);
    const int want[9] = {6, 6, 6, 6, 7, 7, 7, 7, 7};
    for (int i = 0; i < 9; i++)
      if (lst[i]->clusterId != want[i] || !lst[i]->isClassed) { std::printf("FAIL label %d\n", i); return 1; }
    if (db.clusterAmount != 7 || DBImproved::iritatorNum != 99 || lst[4]->isKeyPoint) { std::printf("FAIL counters\n"); return 1; }
    // ICP: data = model rotated 0.5 degrees about z and shifted; go_hell_ICP must undo it
    std::vector<std::unique_ptr<Point3D>> mo, da;
    std::vector<Point3D*> model, data;
    const double t = 0.5 * M_PI / 180.0, c = std::cos(t), s = std::sin(t);
    for (int i = 0; i < 40; i++) {
      double x = (i * 37 % 40) * 3.0, y = (i * 11 % 40) * 2.0, z = (i * 7 % 40) * 1.0;
      mo.emplace_back(new Point3D(x, y, z));
      model.push_back(mo.back().get());
      da.emplace_back(new Point3D(c * x - s * y + 0.25, s * x + c * y - 0.5, z + 0.125));
      data.push_back(da.back().get());
    }
    Matrix R(3, 3), T(3, 1);
    ICP icp(ctx);
    icp.go_hell_ICP(model, data, R, T, 1e-6);
    // inverse of (Rz(t), shift): R = Rz(-t), T = -Rz(-t) * shift
    if (std::fabs(R(0, 0) - c) > 1e-6 || std::fabs(R(0, 1) - s) > 1e-6 || std::fabs(R(2, 2) - 1) > 1e-6) { std::printf("FAIL icp R\n"); return 1; }
    std::printf("PASS host_demo: DBImproved known answer, cf preset, iritatorNum, go_hell_ICP (%d rounds)\n", icp.last_iters);
    return 0;
  } catch (const VcpException& e) {
    std::printf("FAIL exception %d: %s\n", e.code, e.what());
    return 2;
  }
}
