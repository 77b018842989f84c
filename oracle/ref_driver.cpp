// TEST INFRASTRUCTURE ONLY -- never linked into the product (libartp.so).
//
// Drives the REAL patched ODE (compiled by oracle/Makefile from /root/reference/ode, outputs in
// oracle/_ref/) exactly the way art_planner's HeightMapBoxChecker does:
//   ctor            art_planner/src/validity_checker/height_map_box_checker.cpp:11-26
//   setHeightField  art_planner/src/validity_checker/height_map_box_checker.cpp:38-54
//   checkCollision  art_planner/src/validity_checker/height_map_box_checker.cpp:58-72
// The only thing that is not the reference's own code is the Eigen expression
// `layer.rowwise().reverse()` (Eigen is not installed), which is spelled out as a loop below.
//
// Exposed with a C ABI so that tests / fixture generators can reach it through ctypes.
#include <ode/ode.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct RefChecker {
  std::vector<float> mat;  // column-major, rows = size.x  (Eigen::MatrixXf storage order)
  dHeightfieldDataID data;
  dReal rot[12];
  dWorldID world;
  dSpaceID space;
  dBodyID body_field;
  dGeomID geom_field;
  dBodyID body_box;
  dGeomID geom_box;
  dContactGeom contact;
};

}  // namespace

extern "C" {

void* artp_ref_create(float lx, float ly, float lz) {
  RefChecker* c = new RefChecker();
  dInitODE();
  c->world = dWorldCreate();
  c->space = dHashSpaceCreate(0);
  c->body_box = dBodyCreate(c->world);
  c->body_field = dBodyCreate(c->world);
  c->geom_box = dCreateBox(c->space, lx, ly, lz);
  c->data = dGeomHeightfieldDataCreate();
  c->geom_field = dCreateHeightfield(c->space, c->data, 1);
  dRFrom2Axes(c->rot, -1, 0, 0, 0, 0, 1);
  dGeomSetBody(c->geom_box, c->body_box);
  dGeomSetBody(c->geom_field, c->body_field);
  dBodySetRotation(c->body_field, c->rot);
  return c;
}

void artp_ref_destroy(void* h) {
  RefChecker* c = static_cast<RefChecker*>(h);
  dSpaceDestroy(c->space);
  dWorldDestroy(c->world);
  dGeomHeightfieldDataDestroy(c->data);
  dCloseODE();
  delete c;
}

// layer: grid_map layer storage = Eigen::MatrixXf, column-major, rows x cols (rows = size.x).
void artp_ref_set_field(void* h, const float* layer, int rows, int cols, double len_x, double len_y,
                        double pos_x, double pos_y) {
  RefChecker* c = static_cast<RefChecker*>(h);
  c->mat.resize(static_cast<size_t>(rows) * cols);
  // field_.mat = layer.rowwise().reverse()   =>   mat(i, j) = layer(i, cols-1-j)
  float mn = std::numeric_limits<float>::infinity();
  float mx = -std::numeric_limits<float>::infinity();
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) {
      const float v = layer[static_cast<size_t>(i) + static_cast<size_t>(cols - 1 - j) * rows];
      c->mat[static_cast<size_t>(i) + static_cast<size_t>(j) * rows] = v;
      if (std::isfinite(v)) {  // minCoeffOfFinites / maxCoeffOfFinites
        if (v < mn) mn = v;
        if (v > mx) mx = v;
      }
    }
  dGeomHeightfieldDataBuildSingle(c->data, c->mat.data(), 0, len_x, len_y, rows, cols, 1, 0, 0, 0);
  dGeomHeightfieldDataSetBounds(c->data, mn, mx);
  dGeomHeightfieldSetHeightfieldData(c->geom_field, c->data);
  dBodySetPosition(c->body_field, pos_x, pos_y, 0);
}

// poses: n x 16 floats = dPose{origin[4], rotation[12]} (height_map_box_checker.h:20-25).
// hit[i] = (dCollide(box, field, 1, ...) != 0).  Returns the number of poses in contact, like
// HeightMapBoxChecker::checkCollision.
int artp_ref_check(void* h, const float* poses, size_t n, uint8_t* hit) {
  RefChecker* c = static_cast<RefChecker*>(h);
  int n_manifold_with_contact = 0;
  for (size_t i = 0; i < n; ++i) {
    const float* p = poses + 16 * i;
    dBodySetPosition(c->body_box, p[0], p[1], p[2]);
    dBodySetRotation(c->body_box, p + 4);
    const int n_col = dCollide(c->geom_box, c->geom_field, 1, &c->contact, sizeof(dContactGeom));
    if (hit) hit[i] = n_col ? 1 : 0;
    if (n_col) ++n_manifold_with_contact;
  }
  return n_manifold_with_contact;
}

// Per-thread ODE data for multi-threaded timing.
int artp_ref_thread_init(void) { return dAllocateODEDataForThread(dAllocateMaskAll); }

}  // extern "C"
