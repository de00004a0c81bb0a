// On-device occupancy labelling of query points against a triangle mesh, gfx950.
//
// Replaces data_processing/libmesh/inside_mesh.py:5-155 (check_mesh_contains / MeshIntersector /
// TriangleIntersector2d) and the reference's only native component, the Cython TriangleHash
// (data_processing/libmesh/triangle_hash.pyx:8-85), which sit INSIDE the training step when subsample_points > 0
// (trainer/trainer_scene_net.py:112,128 -> mesh_occupancies.py:24-53): there every step copies the projected point
// cloud to the host, rebuilds the hash in Python/Cython and runs numpy over every (point, candidate triangle) pair.
//
// Here: the 2-D triangle hash is built on the host in C++ (same cells, same clamping, triangles in index order) as a
// CSR table, and ONE kernel does the rest per point in float64 with the reference's operation order and one rounding
// per operation (build with -ffp-contract=off): rescale, AABB cull, cell lookup, exact 2-D point-in-triangle test
// (strict inequalities), intersection depth on the z ray, parity in both directions -> contains = odd & odd,
// hole = odd ^ odd.  The booleans equal the reference's bit for bit (tests/golden/mesh_*.npz).
// Points stay on the device (float32 as the trainer produces them, or float64 for the rotated re-tests).
#include "common.h"
#include <algorithm>
#include <cmath>

using namespace svr;

namespace {

struct MeshXf {
  double s[3], t[3];
};

__global__ __launch_bounds__(256) void mesh_contains_kernel(const void *__restrict__ points, int is_f64, int64_t n,
                                                            const double *__restrict__ tri, const int32_t *__restrict__ cell_start,
                                                            const int32_t *__restrict__ tri_ids, int res, MeshXf xf,
                                                            uint8_t *__restrict__ contains, uint8_t *__restrict__ holes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double p[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double raw = is_f64 ? ((const double *)points)[i * 3 + a] : (double)((const float *)points)[i * 3 + a];
    p[a] = xf.s[a] * raw + xf.t[a];  // inside_mesh.py:108-110 (multiply, then add)
  }
  const double r = (double)res;
  const bool inside = 0.0 <= p[0] && p[0] <= r && 0.0 <= p[1] && p[1] <= r && 0.0 <= p[2] && p[2] <= r;  // :41-42 (NaN -> false)
  int n0 = 0, n1 = 0;
  if (inside) {
    const int x = (int)p[0], y = (int)p[1];  // triangle_hash.pyx:65-66
    if (x >= 0 && x < res && y >= 0 && y < res) {
      const int cell = res * x + y;
      for (int k = cell_start[cell]; k < cell_start[cell + 1]; ++k) {
        const double *t = tri + (int64_t)tri_ids[k] * 9;
        const double t0x = t[0], t0y = t[1], t0z = t[2], t1x = t[3], t1y = t[4], t1z = t[5], t2x = t[6], t2y = t[7], t2z = t[8];
        // check_triangles, inside_mesh.py:130-155
        const double A00 = t0x - t2x, A01 = t1x - t2x, A10 = t0y - t2y, A11 = t1y - t2y;
        const double y0 = p[0] - t2x, y1 = p[1] - t2y;
        const double detA = A00 * A11 - A01 * A10;
        const double adet = fabs(detA);
        if (!(adet != 0.0)) continue;
        const double sd = detA > 0.0 ? 1.0 : -1.0;
        const double u = (A11 * y0 - A01 * y1) * sd;
        const double v = ((-A10) * y0 + A00 * y1) * sd;
        const double suv = u + v;
        if (!(0.0 < u && u < adet && 0.0 < v && v < adet && 0.0 < suv && suv < adet)) continue;
        // compute_intersection_depth, inside_mesh.py:76-106 (v1 = t3 - t1, v2 = t2 - t1 in the reference's 1-based names)
        const double v1x = t2x - t0x, v1y = t2y - t0y, v1z = t2z - t0z;
        const double v2x = t1x - t0x, v2y = t1y - t0y, v2z = t1z - t0z;
        const double nx = v1y * v2z - v1z * v2y, ny = v1z * v2x - v1x * v2z, nz = v1x * v2y - v1y * v2x;
        const double alpha = nx * (t0x - p[0]) + ny * (t0y - p[1]);
        const double anz = fabs(nz);
        if (!(anz != 0.0)) continue;  // depth is NaN: counted in neither direction (:62-63)
        const double snz = nz > 0.0 ? 1.0 : -1.0;
        const double depth = t0z * anz + alpha * snz;
        const double pz = p[2] * anz;
        if (depth >= pz) ++n0;
        else if (depth < pz) ++n1;
      }
    }
  }
  const bool c1 = (n0 & 1) != 0, c2 = (n1 & 1) != 0;
  contains[i] = (uint8_t)(c1 && c2);
  holes[i] = (uint8_t)(c1 != c2);
}

// bbox of the vertices the faces reference -> scale / translate (inside_mesh.py:17-22)
bool mesh_transform(const double *verts, int64_t n_verts, const int32_t *faces, int64_t n_faces, int res, MeshXf &xf) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t f = 0; f < n_faces * 3; ++f) {
    const int64_t v = faces[f];
    if (v < 0 || v >= n_verts) return false;
    for (int a = 0; a < 3; ++a) {
      const double c = verts[v * 3 + a];
      lo[a] = std::min(lo[a], c);
      hi[a] = std::max(hi[a], c);
    }
  }
  for (int a = 0; a < 3; ++a) {
    xf.s[a] = (double)(res - 1) / (hi[a] - lo[a]);
    xf.t[a] = 0.5 - xf.s[a] * lo[a];
  }
  return true;
}

inline int clampi(int v, int lo, int hi) { return std::min(std::max(v, lo), hi); }

// integer cell bounding box of a rescaled triangle (triangle_hash.pyx:32-40: <int> truncation, then clamp)
inline void tri_cells(const double *t, int res, int &x0, int &x1, int &y0, int &y1) {
  x0 = clampi((int)std::min(t[0], std::min(t[3], t[6])), 0, res - 1);
  x1 = clampi((int)std::max(t[0], std::max(t[3], t[6])), 0, res - 1);
  y0 = clampi((int)std::min(t[1], std::min(t[4], t[7])), 0, res - 1);
  y1 = clampi((int)std::max(t[1], std::max(t[4], t[7])), 0, res - 1);
}

}  // namespace

extern "C" int64_t svr_mesh_hash_entries(const double *verts, int64_t n_verts, const int32_t *faces, int64_t n_faces,
                                         int32_t res, double *scale_translate) {
  if (!verts || !faces || !scale_translate || n_faces <= 0 || res < 2) {
    set_error("mesh_hash_entries: bad argument (n_faces=%ld res=%d)", (long)n_faces, res);
    return SVR_E_BADARG;
  }
  MeshXf xf;
  if (!mesh_transform(verts, n_verts, faces, n_faces, res, xf)) {
    set_error("mesh_hash_entries: face index out of range");
    return SVR_E_BADARG;
  }
  int64_t entries = 0;
  for (int64_t f = 0; f < n_faces; ++f) {
    double t[9];
    for (int c = 0; c < 3; ++c)
      for (int a = 0; a < 3; ++a) t[c * 3 + a] = xf.s[a] * verts[(int64_t)faces[f * 3 + c] * 3 + a] + xf.t[a];
    int x0, x1, y0, y1;
    tri_cells(t, res, x0, x1, y0, y1);
    entries += (int64_t)(x1 - x0 + 1) * (y1 - y0 + 1);
  }
  for (int a = 0; a < 3; ++a) {
    scale_translate[a] = xf.s[a];
    scale_translate[3 + a] = xf.t[a];
  }
  return entries;
}

extern "C" int svr_mesh_hash_build(const double *verts, int64_t n_verts, const int32_t *faces, int64_t n_faces, int32_t res,
                                   double *tri, int32_t *cell_start, int32_t *tri_ids, int64_t entries) {
  SVR_CHECK(verts && faces && tri && cell_start && tri_ids && n_faces > 0 && res >= 2, SVR_E_BADARG, "mesh_hash_build: bad argument");
  SVR_CHECK(entries < (1LL << 31), SVR_E_UNSUPPORTED, "mesh_hash_build: %ld hash entries", (long)entries);
  MeshXf xf;
  SVR_CHECK(mesh_transform(verts, n_verts, faces, n_faces, res, xf), SVR_E_BADARG, "mesh_hash_build: face index out of range");
  const int64_t cells = (int64_t)res * res;
  std::fill(cell_start, cell_start + cells + 1, 0);
  for (int64_t f = 0; f < n_faces; ++f) {
    double *t = tri + f * 9;
    for (int c = 0; c < 3; ++c)
      for (int a = 0; a < 3; ++a) t[c * 3 + a] = xf.s[a] * verts[(int64_t)faces[f * 3 + c] * 3 + a] + xf.t[a];
    int x0, x1, y0, y1;
    tri_cells(t, res, x0, x1, y0, y1);
    for (int x = x0; x <= x1; ++x)
      for (int y = y0; y <= y1; ++y) ++cell_start[(int64_t)res * x + y + 1];
  }
  for (int64_t c = 0; c < cells; ++c) cell_start[c + 1] += cell_start[c];
  SVR_CHECK(cell_start[cells] == entries, SVR_E_BADARG, "mesh_hash_build: %d entries, caller sized for %ld", cell_start[cells], (long)entries);
  // second pass in triangle order with a moving cursor per cell (the cursor array is rebuilt into cell_start afterwards)
  for (int64_t f = 0; f < n_faces; ++f) {
    int x0, x1, y0, y1;
    tri_cells(tri + f * 9, res, x0, x1, y0, y1);
    for (int x = x0; x <= x1; ++x)
      for (int y = y0; y <= y1; ++y) tri_ids[cell_start[(int64_t)res * x + y]++] = (int32_t)f;
  }
  for (int64_t c = cells; c > 0; --c) cell_start[c] = cell_start[c - 1];
  cell_start[0] = 0;
  return SVR_OK;
}

extern "C" int svr_mesh_contains(const void *points, int32_t points_f64, int64_t n, const double *tri, const int32_t *cell_start,
                                 const int32_t *tri_ids, int32_t res, const double *scale_translate, uint8_t *contains,
                                 uint8_t *holes, void *stream) {
  if (n <= 0) return SVR_OK;
  SVR_CHECK(points && tri && cell_start && tri_ids && scale_translate && contains && holes && res >= 2, SVR_E_BADARG,
            "mesh_contains: bad argument");
  MeshXf xf;
  for (int a = 0; a < 3; ++a) {
    xf.s[a] = scale_translate[a];
    xf.t[a] = scale_translate[3 + a];
  }
  hipLaunchKernelGGL(mesh_contains_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, points, points_f64, n,
                     tri, cell_start, tri_ids, res, xf, contains, holes);
  return launch_status("mesh_contains");
}
