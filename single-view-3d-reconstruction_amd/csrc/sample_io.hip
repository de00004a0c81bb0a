// Sample wire formats of the reference's datasets, read natively, gfx950.
//
// The reference loads every sample in Python (dataset/implicit_dataset.py:24-56): np.load of a zlib-compressed float64
// grid (depth_grid.npz, key 'grid', data_processing/process_sample.py:19-22), read_df (data_processing/
// volume_reader.py:36-45: 3 x uint64 dims + float32 payload, x fastest, unpacked one float at a time through
// struct.unpack into a tuple), and two occupancy_<sigma>.npz files (keys points / occupancies / grid_coords,
// process_sample.py:28-30) whose random subset is assembled through Python lists.  At ~22 ms per training step that
// loader is the next bottleneck.  Here:
//   host (plain C++ + zlib): .df header / payload with one fread into a caller buffer (pinned memory), .npz member
//     lookup (zip central directory, stored or deflated members, zip64 local headers as numpy writes them) and .npy
//     header parsing, straight into a caller buffer;
//   device: x-fastest -> C-order transpose of the .df payload, float64 -> float32 casts, and the per-sample random
//     row subset (gather by index with cast: points float64 -> float32, occupancies bool -> float32).
// Given the same files and the same indices the tensors equal the reference's __getitem__ outputs bit for bit.
#include "common.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

using namespace svr;

namespace {

struct File {
  FILE *f = nullptr;
  explicit File(const char *path) { f = fopen(path, "rb"); }
  ~File() { if (f) fclose(f); }
};

uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64(const unsigned char *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

struct ZipEntry {
  int method = -1;
  uint64_t comp = 0, uncomp = 0, local = 0;
};

// locate `name` in the zip's central directory (zip64 end record / extra fields handled)
int zip_find(FILE *f, const std::string &name, ZipEntry &e) {
  if (fseek(f, 0, SEEK_END)) return -1;
  const long size = ftell(f);
  const long tail = size < 65557 ? size : 65557;
  std::vector<unsigned char> buf((size_t)tail);
  if (fseek(f, size - tail, SEEK_SET) || fread(buf.data(), 1, (size_t)tail, f) != (size_t)tail) return -1;
  long eocd = -1;
  for (long i = tail - 22; i >= 0; --i)
    if (rd32(&buf[i]) == 0x06054b50u) { eocd = i; break; }
  if (eocd < 0) return -1;
  uint64_t n = rd16(&buf[eocd + 10]), cd_off = rd32(&buf[eocd + 16]), cd_size = rd32(&buf[eocd + 12]);
  if ((n == 0xFFFF || cd_off == 0xFFFFFFFFu || cd_size == 0xFFFFFFFFu) && eocd >= 20 && rd32(&buf[eocd - 20]) == 0x07064b50u) {
    const uint64_t z64 = rd64(&buf[eocd - 20 + 8]);  // zip64 end-of-central-directory record
    unsigned char r[56];
    if (fseek(f, (long)z64, SEEK_SET) || fread(r, 1, 56, f) != 56 || rd32(r) != 0x06064b50u) return -1;
    n = rd64(r + 32);
    cd_size = rd64(r + 40);
    cd_off = rd64(r + 48);
  }
  // a truncated or garbage file must not size an allocation or a seek: the directory lies inside the file
  if (cd_off > (uint64_t)size || cd_size > (uint64_t)size - cd_off) return -1;
  std::vector<unsigned char> cd((size_t)cd_size);
  if (fseek(f, (long)cd_off, SEEK_SET) || fread(cd.data(), 1, cd.size(), f) != cd.size()) return -1;
  size_t p = 0;
  for (uint64_t i = 0; i < n && p + 46 <= cd.size(); ++i) {
    if (rd32(&cd[p]) != 0x02014b50u) return -1;
    const int method = rd16(&cd[p + 10]);
    uint64_t comp = rd32(&cd[p + 20]), uncomp = rd32(&cd[p + 24]), local = rd32(&cd[p + 42]);
    const size_t nl = rd16(&cd[p + 28]), xl = rd16(&cd[p + 30]), cl = rd16(&cd[p + 32]);
    if (p + 46 + nl + xl + cl > cd.size()) return -1;  // entry runs past the directory
    const std::string fn((const char *)&cd[p + 46], nl);
    size_t x = p + 46 + nl;
    const size_t xend = x + xl;
    while (x + 4 <= xend) {  // zip64 extended information: only the fields that overflowed, in this order
      const int id = rd16(&cd[x]), sz = rd16(&cd[x + 2]);
      if (id == 1) {
        size_t q = x + 4;
        const size_t fend = x + 4 + (size_t)sz < xend ? x + 4 + (size_t)sz : xend;
        if (uncomp == 0xFFFFFFFFu) { if (q + 8 > fend) return -1; uncomp = rd64(&cd[q]); q += 8; }
        if (comp == 0xFFFFFFFFu) { if (q + 8 > fend) return -1; comp = rd64(&cd[q]); q += 8; }
        if (local == 0xFFFFFFFFu) { if (q + 8 > fend) return -1; local = rd64(&cd[q]); q += 8; }
      }
      x += 4 + (size_t)sz;
    }
    if (fn == name) {
      e.method = method;
      e.comp = comp;
      e.uncomp = uncomp;
      e.local = local;
      return 0;
    }
    p += 46 + nl + xl + cl;
  }
  return 1;  // not found
}

// read the member's bytes [skip, skip + want) of its UNCOMPRESSED stream into out
int zip_read(FILE *f, const ZipEntry &e, uint64_t skip, void *out, uint64_t want) {
  unsigned char lh[30];
  if (fseek(f, (long)e.local, SEEK_SET) || fread(lh, 1, 30, f) != 30 || rd32(lh) != 0x04034b50u) return -1;
  const long data = (long)e.local + 30 + rd16(lh + 26) + rd16(lh + 28);
  if (skip + want > e.uncomp) return -1;
  if (e.method == 0) {
    if (fseek(f, data + (long)skip, SEEK_SET) || fread(out, 1, (size_t)want, f) != (size_t)want) return -1;
    return 0;
  }
  if (e.method != 8) return -2;
  if (fseek(f, data, SEEK_SET)) return -1;
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) return -1;
  std::vector<unsigned char> in(1 << 16), scratch(1 << 16);
  uint64_t left_in = e.comp, produced = 0;
  int rc = Z_OK;
  while (produced < skip + want && rc != Z_STREAM_END) {
    if (zs.avail_in == 0 && left_in > 0) {
      const size_t take = (size_t)(left_in < in.size() ? left_in : in.size());
      if (fread(in.data(), 1, take, f) != take) { inflateEnd(&zs); return -1; }
      zs.next_in = in.data();
      zs.avail_in = (uInt)take;
      left_in -= take;
    }
    if (produced < skip) {  // header bytes in front of the payload: inflate into a scratch buffer
      const uint64_t room = skip - produced;
      zs.next_out = scratch.data();
      zs.avail_out = (uInt)(room < scratch.size() ? room : scratch.size());
    } else {
      const uint64_t room = skip + want - produced;
      zs.next_out = (unsigned char *)out + (produced - skip);
      zs.avail_out = (uInt)(room < (1u << 30) ? room : (1u << 30));
    }
    const uInt before = zs.avail_out;
    rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); return -1; }
    produced += before - zs.avail_out;
    if (before == zs.avail_out && zs.avail_in == 0 && left_in == 0) break;
  }
  inflateEnd(&zs);
  return produced >= skip + want ? 0 : -1;
}

struct NpyInfo {
  int dtype = -1;  // SVR_DT_*
  int ndim = 0, fortran = 0;
  int64_t shape[8] = {0};
  uint64_t header_bytes = 0;
};

int dtype_code(const std::string &descr) {
  if (descr == "<f4" || descr == "=f4") return SVR_DT_F32;
  if (descr == "<f8" || descr == "=f8") return SVR_DT_F64;
  if (descr == "|b1") return SVR_DT_BOOL;
  if (descr == "|u1") return SVR_DT_U8;
  if (descr == "<i4" || descr == "=i4") return SVR_DT_I32;
  if (descr == "<i8" || descr == "=i8") return SVR_DT_I64;
  return -1;
}

int dtype_size(int code) {
  switch (code) {
    case SVR_DT_F32: case SVR_DT_I32: return 4;
    case SVR_DT_F64: case SVR_DT_I64: return 8;
    case SVR_DT_BOOL: case SVR_DT_U8: return 1;
  }
  return 0;
}

// .npy header: magic, version, little-endian header length, python dict literal
int npy_parse(FILE *f, const ZipEntry &e, NpyInfo &info) {
  unsigned char h[12];
  const uint64_t first = e.uncomp < 12 ? e.uncomp : 12;
  if (first < 10 || zip_read(f, e, 0, h, first)) return -1;
  if (memcmp(h, "\x93NUMPY", 6) != 0) return -1;
  uint64_t hlen, pre;
  if (h[6] == 1) { hlen = rd16(h + 8); pre = 10; } else { if (first < 12) return -1; hlen = rd32(h + 8); pre = 12; }
  if (pre + hlen > e.uncomp || hlen > (1u << 20)) return -1;  // (numpy headers are a few hundred bytes)
  std::string d((size_t)hlen, '\0');
  if (zip_read(f, e, pre, &d[0], hlen)) return -1;
  info.header_bytes = pre + hlen;
  auto value_after = [&](const char *key) -> size_t {
    size_t k = d.find(key);
    if (k == std::string::npos) return k;
    k = d.find(':', k);
    return k == std::string::npos ? k : k + 1;
  };
  size_t p = value_after("'descr'");
  if (p == std::string::npos) return -1;
  size_t q0 = d.find('\'', p), q1 = q0 == std::string::npos ? q0 : d.find('\'', q0 + 1);
  if (q1 == std::string::npos) return -1;
  info.dtype = dtype_code(d.substr(q0 + 1, q1 - q0 - 1));
  p = value_after("'fortran_order'");
  if (p == std::string::npos) return -1;
  const size_t fo = d.find_first_not_of(' ', p);
  if (fo == std::string::npos) return -1;
  info.fortran = d.compare(fo, 4, "True") == 0;
  p = value_after("'shape'");
  if (p == std::string::npos) return -1;
  size_t a = d.find('(', p), b = d.find(')', a);
  if (a == std::string::npos || b == std::string::npos) return -1;
  info.ndim = 0;
  const char *s = d.c_str() + a + 1, *end = d.c_str() + b;
  while (s < end && info.ndim < 8) {
    while (s < end && (*s == ' ' || *s == ',')) ++s;
    if (s >= end) break;
    char *nx;
    info.shape[info.ndim++] = strtoll(s, &nx, 10);
    if (nx == s) return -1;
    s = nx;
  }
  return 0;
}

int open_member(const char *path, const char *member, File &fh, ZipEntry &e, NpyInfo &info, const char *what) {
  SVR_CHECK(path && member, SVR_E_BADARG, "%s: null argument", what);
  SVR_CHECK(fh.f != nullptr, SVR_E_IO, "%s: cannot open %s", what, path);
  int rc = zip_find(fh.f, std::string(member) + ".npy", e);
  SVR_CHECK(rc == 0, rc > 0 ? SVR_E_NOTFOUND : SVR_E_IO, "%s: %s has no member '%s'", what, path, member);
  SVR_CHECK(npy_parse(fh.f, e, info) == 0, SVR_E_IO, "%s: %s[%s]: malformed .npy header", what, path, member);
  SVR_CHECK(info.dtype >= 0, SVR_E_UNSUPPORTED, "%s: %s[%s]: unsupported dtype", what, path, member);
  return SVR_OK;
}

// ---- device side ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void df_to_grid_kernel(const float *__restrict__ payload, float *__restrict__ out, int X, int Y, int Z) {
  // out[x][y][z] (C order) = payload[x + X * (y + Y * z)]; 32 x 32 tiles through LDS over the (x, z) plane of one y
  __shared__ float tile[32][33];
  const int y = blockIdx.z, x0 = blockIdx.x * 32, z0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int x = x0 + tx, z = z0 + r;
    tile[r][tx] = (x < X && z < Z) ? payload[x + (int64_t)X * (y + (int64_t)Y * z)] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int x = x0 + r, z = z0 + tx;
    if (x < X && z < Z) out[((int64_t)x * Y + y) * Z + z] = tile[tx][r];
  }
}

template <typename T>
__device__ __forceinline__ float as_f32(T v) { return (float)v; }

template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T *__restrict__ in, float *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = as_f32(in[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void subsample_rows_kernel(const T *__restrict__ rows, int64_t n_rows, int cols,
                                                             const int64_t *__restrict__ idx, int64_t n_idx, float *__restrict__ out,
                                                             int *__restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_idx * cols) return;
  const int64_t r = idx[i / cols];
  if (r < 0 || r >= n_rows) {
    if (bad) atomicExch(bad, 1);
    out[i] = 0.f;
    return;
  }
  out[i] = as_f32(rows[r * cols + i % cols]);
}

}  // namespace

extern "C" int svr_df_dims(const char *path, int64_t *dims) {
  SVR_CHECK(path && dims, SVR_E_BADARG, "df_dims: null argument");
  File fh(path);
  SVR_CHECK(fh.f != nullptr, SVR_E_IO, "df_dims: cannot open %s", path);
  unsigned char h[24];
  SVR_CHECK(fread(h, 1, 24, fh.f) == 24, SVR_E_IO, "df_dims: %s: short header", path);
  for (int a = 0; a < 3; ++a) dims[a] = (int64_t)rd64(h + 8 * a);
  SVR_CHECK(dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && dims[0] < (1 << 20) && dims[1] < (1 << 20) && dims[2] < (1 << 20), SVR_E_IO,
            "df_dims: %s: implausible dims %ld x %ld x %ld", path, (long)dims[0], (long)dims[1], (long)dims[2]);
  return SVR_OK;
}

extern "C" int svr_df_read(const char *path, float *payload, int64_t n) {
  int64_t dims[3];
  if (int rc = svr_df_dims(path, dims)) return rc;
  SVR_CHECK(payload && n == dims[0] * dims[1] * dims[2], SVR_E_BADSHAPE, "df_read: %s holds %ld values, caller sized for %ld", path,
            (long)(dims[0] * dims[1] * dims[2]), (long)n);
  File fh(path);
  SVR_CHECK(fh.f != nullptr && fseek(fh.f, 24, SEEK_SET) == 0, SVR_E_IO, "df_read: cannot open %s", path);
  SVR_CHECK(fread(payload, sizeof(float), (size_t)n, fh.f) == (size_t)n, SVR_E_IO, "df_read: %s: truncated payload", path);  // like the reference's `raise Exception`
  return SVR_OK;
}

// The host readers never throw across the ABI: std::bad_alloc / std::length_error / std::out_of_range from a malformed file
// come back as SVR_E_IO (a half-written occupancy_*.npz must not abort the training process).
namespace {

int npz_member_info_impl(const char *path, const char *member, int32_t *dtype, int32_t *ndim, int64_t *shape, int32_t *fortran_order) {
  File fh(path);
  ZipEntry e;
  NpyInfo info;
  if (int rc = open_member(path, member, fh, e, info, "npz_member_info")) return rc;
  SVR_CHECK(dtype && ndim && shape && fortran_order, SVR_E_BADARG, "npz_member_info: null output");
  *dtype = info.dtype;
  *ndim = info.ndim;
  *fortran_order = info.fortran;
  for (int i = 0; i < 8; ++i) shape[i] = i < info.ndim ? info.shape[i] : 0;
  return SVR_OK;
}

int npz_member_read_impl(const char *path, const char *member, void *out, int64_t nbytes) {
  File fh(path);
  ZipEntry e;
  NpyInfo info;
  if (int rc = open_member(path, member, fh, e, info, "npz_member_read")) return rc;
  int64_t count = 1;
  for (int i = 0; i < info.ndim; ++i) {
    SVR_CHECK(info.shape[i] >= 0 && (info.shape[i] == 0 || count <= (int64_t)(1LL << 56) / info.shape[i]), SVR_E_IO,
              "npz_member_read: %s[%s]: implausible shape", path, member);
    count *= info.shape[i];
  }
  SVR_CHECK(out && nbytes == count * dtype_size(info.dtype), SVR_E_BADSHAPE, "npz_member_read: %s[%s] holds %ld bytes, caller sized for %ld",
            path, member, (long)(count * dtype_size(info.dtype)), (long)nbytes);
  int rc = zip_read(fh.f, e, info.header_bytes, out, (uint64_t)nbytes);
  SVR_CHECK(rc == 0, rc == -2 ? SVR_E_UNSUPPORTED : SVR_E_IO, "npz_member_read: %s[%s]: %s", path, member,
            rc == -2 ? "unsupported zip compression method" : "read / inflate failed");
  return SVR_OK;
}

template <typename F>
int io_guard(const char *what, F &&body) {
  try {
    return body();
  } catch (const std::exception &ex) {
    SVR_CHECK(false, SVR_E_IO, "%s: malformed file (%s)", what, ex.what());
  } catch (...) {
    SVR_CHECK(false, SVR_E_IO, "%s: malformed file", what);
  }
  return SVR_E_IO;
}

}  // namespace

extern "C" int svr_npz_member_info(const char *path, const char *member, int32_t *dtype, int32_t *ndim, int64_t *shape,
                                   int32_t *fortran_order) {
  return io_guard("npz_member_info", [&] { return npz_member_info_impl(path, member, dtype, ndim, shape, fortran_order); });
}

extern "C" int svr_npz_member_read(const char *path, const char *member, void *out, int64_t nbytes) {
  return io_guard("npz_member_read", [&] { return npz_member_read_impl(path, member, out, nbytes); });
}

extern "C" int svr_df_to_grid(const float *payload, float *out, int32_t X, int32_t Y, int32_t Z, void *stream) {
  SVR_CHECK(payload && out && X > 0 && Y > 0 && Z > 0 && Y < 65536 && cdiv(Z, 32) < 65536, SVR_E_BADARG, "df_to_grid: bad argument");
  hipLaunchKernelGGL(df_to_grid_kernel, dim3((unsigned)cdiv(X, 32), (unsigned)cdiv(Z, 32), (unsigned)Y), dim3(256), 0,
                     (hipStream_t)stream, payload, out, X, Y, Z);
  return launch_status("df_to_grid");
}

extern "C" int svr_cast_to_f32(const void *in, int32_t dtype, float *out, int64_t n, void *stream) {
  if (n <= 0) return SVR_OK;
  SVR_CHECK(in && out, SVR_E_BADARG, "cast_to_f32: null pointer");
  const dim3 g((unsigned)cdiv(n, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case SVR_DT_F64: hipLaunchKernelGGL(cast_to_f32_kernel<double>, g, b, 0, s, (const double *)in, out, n); break;
    case SVR_DT_F32: hipLaunchKernelGGL(cast_to_f32_kernel<float>, g, b, 0, s, (const float *)in, out, n); break;
    case SVR_DT_BOOL: case SVR_DT_U8: hipLaunchKernelGGL(cast_to_f32_kernel<uint8_t>, g, b, 0, s, (const uint8_t *)in, out, n); break;
    case SVR_DT_I32: hipLaunchKernelGGL(cast_to_f32_kernel<int32_t>, g, b, 0, s, (const int32_t *)in, out, n); break;
    case SVR_DT_I64: hipLaunchKernelGGL(cast_to_f32_kernel<int64_t>, g, b, 0, s, (const int64_t *)in, out, n); break;
    default: SVR_CHECK(false, SVR_E_UNSUPPORTED, "cast_to_f32: dtype %d", dtype);
  }
  return launch_status("cast_to_f32");
}

extern "C" int svr_subsample_rows(const void *rows, int32_t dtype, int64_t n_rows, int32_t cols, const int64_t *idx, int64_t n_idx,
                                  float *out, int32_t *bad_flag, void *stream) {
  if (n_idx <= 0) return SVR_OK;
  SVR_CHECK(rows && idx && out && cols > 0 && n_rows > 0, SVR_E_BADARG, "subsample_rows: bad argument");
  const dim3 g((unsigned)cdiv(n_idx * cols, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case SVR_DT_F64: hipLaunchKernelGGL(subsample_rows_kernel<double>, g, b, 0, s, (const double *)rows, n_rows, cols, idx, n_idx, out, bad_flag); break;
    case SVR_DT_F32: hipLaunchKernelGGL(subsample_rows_kernel<float>, g, b, 0, s, (const float *)rows, n_rows, cols, idx, n_idx, out, bad_flag); break;
    case SVR_DT_BOOL: case SVR_DT_U8: hipLaunchKernelGGL(subsample_rows_kernel<uint8_t>, g, b, 0, s, (const uint8_t *)rows, n_rows, cols, idx, n_idx, out, bad_flag); break;
    default: SVR_CHECK(false, SVR_E_UNSUPPORTED, "subsample_rows: dtype %d", dtype);
  }
  return launch_status("subsample_rows");
}
