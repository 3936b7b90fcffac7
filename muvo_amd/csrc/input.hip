// Input pipeline on the device: lidar sweep -> 64x1024 range view and sparse voxel list -> dense grid,
// the per-frame work of CarlaDataset.load_single_element_time_t (muvo/data/dataset.py:275-327) and
// PointCloud.do_range_projection (muvo/utils/geometry_utils.py:176-213).  HBM-bound scatter kernels; the reference's
// "sort by depth, scatter, last write wins" becomes a 64-bit atomicMin on (depth, point index) per pixel.
#include "common.h"

struct RangeArgs {
  double lidar[3];        // POINTS.LIDAR_POSITION
  double ego_lo[3], ego_hi[3];
  double fov_down_abs, fov;   // |fov_down|, fov_up - fov_down  (radians)
  int H, W;
};

// Geometry of one point exactly as the reference does it: float32 conversion to the ego frame, float64 projection.
__device__ __forceinline__ bool range_point(const float* __restrict__ raw, long i, const RangeArgs& a, float (&p)[3], double& depth,
                                            int& ph, int& pw) {
  // convert_coor_lidar (data_preprocessing.py:119-122): float32 += position, y mirrored
  p[0] = (float)((double)raw[i * 3] + a.lidar[0]);
  p[1] = -(float)((double)raw[i * 3 + 1] + a.lidar[1]);
  p[2] = (float)((double)raw[i * 3 + 2] + a.lidar[2]);
  // ego-vehicle box (dataset.py:286-290), strict inequalities in float64
  const bool ego = a.ego_lo[0] < (double)p[0] && (double)p[0] < a.ego_hi[0] && a.ego_lo[1] < (double)p[1] &&
                   (double)p[1] < a.ego_hi[1] && a.ego_lo[2] < (double)p[2] && (double)p[2] < a.ego_hi[2];
  if (ego) return false;
  // do_range_projection (geometry_utils.py:176-198)
  const double cx = (double)p[0] - a.lidar[0], cy = -(double)p[1] - a.lidar[1], cz = (double)p[2] - a.lidar[2];
  depth = sqrt(cx * cx + cy * cy + cz * cz);
  const double yaw = atan2(-cy, cx), pitch = asin(cz / depth);
  double fw = floor(0.5 * (1.0 - yaw / M_PI) * (double)a.W), fh = floor((1.0 - (pitch + a.fov_down_abs) / a.fov) * (double)a.H);
  fw = fmin((double)(a.W - 1), fw); fw = fmax(0.0, fw);
  fh = fmin((double)(a.H - 1), fh); fh = fmax(0.0, fh);
  pw = (int)fw; ph = (int)fh;
  return true;
}

// pass 1: best[pixel] = min over its points of the depth bit pattern (non-negative doubles order like their bits)
__global__ void __launch_bounds__(256)
range_min_depth_kernel(const float* __restrict__ raw, long P, RangeArgs a, unsigned long long* __restrict__ best) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < P; i += (long)gridDim.x * 256) {
    float p[3]; double depth; int ph, pw;
    if (!range_point(raw, i, a, p, depth, ph, pw)) continue;
    atomicMin(best + (long)ph * a.W + pw, (unsigned long long)__double_as_longlong(depth));
  }
}
// pass 2: among the points at the minimum depth of a pixel the lowest index wins
__global__ void __launch_bounds__(256)
range_min_index_kernel(const float* __restrict__ raw, long P, RangeArgs a, const unsigned long long* __restrict__ best,
                       unsigned int* __restrict__ winner) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < P; i += (long)gridDim.x * 256) {
    float p[3]; double depth; int ph, pw;
    if (!range_point(raw, i, a, p, depth, ph, pw)) continue;
    const long px = (long)ph * a.W + pw;
    if ((unsigned long long)__double_as_longlong(depth) == best[px]) atomicMin(winner + px, (unsigned int)i);
  }
}
// pass 3: one thread per pixel writes x, y, z, depth and the remapped label of the winner (or the fill values)
__global__ void __launch_bounds__(256)
range_write_kernel(const float* __restrict__ raw, const unsigned char* __restrict__ tag, const unsigned char* __restrict__ remap,
                   RangeArgs a, const unsigned int* __restrict__ winner, float* __restrict__ xyzd, unsigned char* __restrict__ seg) {
  const long HW = (long)a.H * a.W;
  for (long px = blockIdx.x * 256L + threadIdx.x; px < HW; px += (long)gridDim.x * 256) {
    const unsigned int w = winner[px];
    float p[3] = {0.f, 0.f, 0.f};
    float d = -1.f;
    unsigned char s = 0;
    if (w != 0xffffffffu) {
      double depth; int ph, pw;
      range_point(raw, (long)w, a, p, depth, ph, pw);
      d = (float)depth;
      s = remap[tag[w]];
    }
    xyzd[px] = p[0]; xyzd[HW + px] = p[1]; xyzd[2 * HW + px] = p[2]; xyzd[3 * HW + px] = d;
    if (seg) seg[px] = s;
  }
}

// voxels[x][y][z] = remap[tag] of the LAST row that names the voxel (numpy fancy assignment): atomicMax on (row << 8 | value)
__global__ void __launch_bounds__(256)
voxel_scatter_kernel(const long long* __restrict__ rows, long Q, const unsigned char* __restrict__ remap, int X, int Y, int Z,
                     unsigned int* __restrict__ key) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < Q; i += (long)gridDim.x * 256) {
    const long long x = rows[i * 4], y = rows[i * 4 + 1], z = rows[i * 4 + 2];
    long long t = rows[i * 4 + 3];
    if (x < 0 || x >= X || y < 0 || y >= Y || z < 0 || z >= Z) continue;
    if (t == 255) t = 0;
    atomicMax(key + ((x * Y + y) * Z + z), ((unsigned int)(i + 1) << 8) | (unsigned int)remap[t]);
  }
}
__global__ void __launch_bounds__(256)
voxel_decode_kernel(const unsigned int* __restrict__ key, unsigned char* __restrict__ vox, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) vox[i] = (unsigned char)(key[i] & 0xffu);
}


// ---- instance ids -> centre heat map and offset labels (muvo/utils/instance_utils.py:4-35) -------------------------------
// pass 1: per (frame, id) pixel count and coordinate sums; pass 2: per pixel the heat-map maximum over the instances present in
// its frame and the offset to the centroid of its own instance.  Centroid = round-half-even of the mean (torch.round).
__global__ void __launch_bounds__(256)
instance_sums_kernel(const unsigned char* __restrict__ inst, long F, int H, int W, double* __restrict__ sums) {
  const long n = F * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int id = inst[i];
    if (id == 0) continue;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long f = i / ((long)W * H);
    double* s = sums + (f * 256 + id) * 3;
    atomicAdd(s, 1.0); atomicAdd(s + 1, (double)y); atomicAdd(s + 2, (double)x);
  }
}
__global__ void __launch_bounds__(256)
instance_labels_kernel(const unsigned char* __restrict__ inst, const double* __restrict__ sums, long F, int H, int W, float sigma,
                       float ignore, float* __restrict__ center, float* __restrict__ offset) {
  __shared__ float cy[256], cx[256];
  __shared__ int present[256], npres;
  const long f = blockIdx.y;
  if (threadIdx.x == 0) npres = 0;
  __syncthreads();
  {
    const int id = threadIdx.x;
    const double* s = sums + (f * 256 + id) * 3;
    if (id > 0 && s[0] > 0.0) {
      // x[instance_mask].mean().round(): float32 mean of exact integer sums, then round-half-even
      cy[id] = rintf((float)s[1] / (float)s[0]);
      cx[id] = rintf((float)s[2] / (float)s[0]);
      present[atomicAdd(&npres, 1)] = id;
    }
  }
  __syncthreads();
  const long HW = (long)H * W;
  const float inv = 1.f / (sigma * sigma);
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const float y = (float)(p / W), x = (float)(p % W);
    float g = 0.f;
    for (int k = 0; k < npres; ++k) {
      const int id = present[k];
      const float oy = cy[id] - y, ox = cx[id] - x;
      g = fmaxf(g, expf(-(oy * oy + ox * ox) * inv));
    }
    center[f * HW + p] = g;
    const int id = inst[f * HW + p];
    offset[(f * 2) * HW + p] = id ? cy[id] - y : ignore;
    offset[(f * 2 + 1) * HW + p] = id ? cx[id] - x : ignore;
  }
}

#define ST ((hipStream_t)stream)
extern "C" {

int muvo_range_projection(const float* points_xyz, const uint8_t* obj_tag, const uint8_t* remap, int64_t P, const double* lidar_pos,
                          const double* ego_dim, double fov_down_deg, double fov_up_deg, int H, int W, void* scratch,
                          float* xyzd, uint8_t* seg, void* stream) {
  MUVO_CHECK_ARG(points_xyz && obj_tag && remap && lidar_pos && ego_dim && scratch && xyzd, "range_projection: null pointer");
  MUVO_CHECK_ARG(P >= 0 && P < 0xffffffffll && H > 0 && W > 0, "range_projection: bad sizes");
  RangeArgs a;
  for (int k = 0; k < 3; ++k) a.lidar[k] = lidar_pos[k];
  a.ego_lo[0] = -ego_dim[0] / 2; a.ego_lo[1] = -ego_dim[1] / 2; a.ego_lo[2] = 0.0;
  a.ego_hi[0] = ego_dim[0] / 2; a.ego_hi[1] = ego_dim[1] / 2; a.ego_hi[2] = ego_dim[2];
  const double fd = fov_down_deg / 180.0 * M_PI, fu = fov_up_deg / 180.0 * M_PI;
  a.fov_down_abs = fabs(fd); a.fov = fu - fd; a.H = H; a.W = W;
  const long HW = (long)H * W;
  unsigned long long* best = (unsigned long long*)scratch;
  unsigned int* winner = (unsigned int*)(best + HW);
  if (hipMemsetAsync(scratch, 0xff, (size_t)HW * 12, ST) != hipSuccess) {
    muvo_set_error("range_projection: memset failed");
    return MUVO_ERR_HIP;
  }
  if (P > 0) {
    hipLaunchKernelGGL(range_min_depth_kernel, dim3(ew_grid(P)), dim3(256), 0, ST, points_xyz, (long)P, a, best);
    hipLaunchKernelGGL(range_min_index_kernel, dim3(ew_grid(P)), dim3(256), 0, ST, points_xyz, (long)P, a, best, winner);
  }
  hipLaunchKernelGGL(range_write_kernel, dim3(ew_grid(HW)), dim3(256), 0, ST, points_xyz, obj_tag, remap, a, winner, xyzd, seg);
  MUVO_CHECK_LAUNCH("range_projection");
  return MUVO_OK;
}

int muvo_voxel_grid(const int64_t* rows, int64_t Q, const uint8_t* remap, int X, int Y, int Z, uint32_t* scratch, uint8_t* voxels,
                    void* stream) {
  MUVO_CHECK_ARG(rows && remap && scratch && voxels && Q >= 0 && Q < (1 << 24) && X > 0 && Y > 0 && Z > 0, "voxel_grid: bad args (Q < 2^24)");
  const long n = (long)X * Y * Z;
  if (hipMemsetAsync(scratch, 0, sizeof(uint32_t) * (size_t)n, ST) != hipSuccess) {
    muvo_set_error("voxel_grid: memset failed");
    return MUVO_ERR_HIP;
  }
  if (Q > 0) hipLaunchKernelGGL(voxel_scatter_kernel, dim3(ew_grid(Q)), dim3(256), 0, ST, (const long long*)rows, (long)Q, remap, X, Y, Z, scratch);
  hipLaunchKernelGGL(voxel_decode_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, scratch, voxels, n);
  MUVO_CHECK_LAUNCH("voxel_grid");
  return MUVO_OK;
}

int muvo_instance_labels(const uint8_t* instance, int64_t F, int H, int W, float sigma, float ignore, double* scratch, float* center,
                         float* offset, void* stream) {
  MUVO_CHECK_ARG(instance && scratch && center && offset && F > 0 && F <= 65535 && H > 0 && W > 0 && sigma > 0.f, "instance_labels: bad args");
  if (hipMemsetAsync(scratch, 0, sizeof(double) * (size_t)F * 256 * 3, ST) != hipSuccess) {
    muvo_set_error("instance_labels: memset failed");
    return MUVO_ERR_HIP;
  }
  hipLaunchKernelGGL(instance_sums_kernel, dim3(ew_grid(F * H * W)), dim3(256), 0, ST, instance, (long)F, H, W, scratch);
  long nb = ((long)H * W + 255) / 256;
  if (nb > 64) nb = 64;
  hipLaunchKernelGGL(instance_labels_kernel, dim3((unsigned)nb, (unsigned)F), dim3(256), 0, ST, instance, scratch, (long)F, H, W, sigma, ignore,
                     center, offset);
  MUVO_CHECK_LAUNCH("instance_labels");
  return MUVO_OK;
}

}  // extern "C"
