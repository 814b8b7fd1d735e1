// Post-sampling transform of sample/generate.py:160-166, fused on the device:
//     sample = inv_transform(sample.cpu().permute(0, 2, 3, 1))          data * std + mean        (dataset.py:132-133)
//     sample = recover_from_ric(sample, 22)                              motion_process.py:437-452, :366-385
//     sample = sample.view(-1, 22, 3 -> ...).permute(0, 2, 3, 1)         -> [B, 22, 3, T]
// The reference does this on the CPU after a D2H copy of the normalised features; here the [B, 263, 1, T] sample stays
// in HBM and one workgroup per motion produces the joint positions.  HBM-bound: reads 67 of the 263 feature rows
// (rotation velocity, root xz velocity, root height, 21 x 3 rotation-invariant coordinates), writes 66 rows.
//   * frames are the contiguous axis of both tensors, one lane per frame: every load/store is coalesced;
//   * the two prefix sums over time (heading angle, root xz) are done by ONE lane in the reference's order -- a 196-step
//     fp32 add chain is ~1 us, and it keeps the result within rounding of torch.cumsum instead of a re-associated scan;
//   * quaternion algebra is written exactly as common/quaternion.py:56-75 (qrot) with q = (cos a, 0, sin a, 0).
#pragma once
#include "common.h"

namespace mdm {

// v' = qrot(qinv(q), v) for q = (c, 0, s, 0): qvec of qinv(q) = (0, -s, 0)   (quaternion.py:16-20, :56-75)
__device__ __forceinline__ void qrot_inv_y(float c, float s, float vx, float vy, float vz, float& ox, float& oy, float& oz) {
  const float qx = 0.f, qy = -s, qz = 0.f;
  const float uvx = qy * vz - qz * vy, uvy = qz * vx - qx * vz, uvz = qx * vy - qy * vx;
  const float uuvx = qy * uvz - qz * uvy, uuvy = qz * uvx - qx * uvz, uuvz = qx * uvy - qy * uvx;
  ox = vx + 2.f * (c * uvx + uuvx);
  oy = vy + 2.f * (c * uvy + uuvy);
  oz = vz + 2.f * (c * uvz + uuvz);
}

// x [B][JF][T] normalised features (JF = 4 + (J-1)*3 + ... >= 4 + 3 (J-1)), mean/std [JF]; out [B][J][3][T].
// Dynamic LDS: 3 * T floats.
__global__ __launch_bounds__(256) void recover_from_ric_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ stdv, float* __restrict__ out,
                                                               int T, int JF, int J) {
  MDM_DYN_SMEM(float, sm);
  float* s_ang = sm;          // rot_vel, then heading angle
  float* s_x = sm + T;        // root x velocity, then rotated, then position
  float* s_z = sm + 2 * T;
  const int b = blockIdx.x;
  const float* xb = x + (size_t)b * JF * T;
  auto feat = [&](int f, int t) { return xb[(size_t)f * T + t] * stdv[f] + mean[f]; };

  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    s_ang[t] = feat(0, t);
    s_x[t] = feat(1, t);
    s_z[t] = feat(2, t);
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // r_rot_ang[1:] = rot_vel[:-1]; cumsum   (motion_process.py:367-371)
    float acc = 0.f, prev = s_ang[0];
    s_ang[0] = 0.f;
    for (int t = 1; t < T; ++t) {
      const float cur = s_ang[t];
      acc += prev;
      s_ang[t] = acc;
      prev = cur;
    }
  }
  __syncthreads();
  // r_pos[1:, [0, 2]] = data[:-1, 1:3]; r_pos = qrot(qinv(r_rot_quat), r_pos)   (:377-380)
  float vx_prev[4], vz_prev[4];   // this lane's frames (blockDim strided), loaded before anyone overwrites s_x / s_z
  int nmine = 0;
  for (int t = threadIdx.x; t < T; t += blockDim.x, ++nmine) {
    if (nmine < 4) {
      vx_prev[nmine] = (t >= 1) ? s_x[t - 1] : 0.f;
      vz_prev[nmine] = (t >= 1) ? s_z[t - 1] : 0.f;
    }
  }
  __syncthreads();
  nmine = 0;
  for (int t = threadIdx.x; t < T; t += blockDim.x, ++nmine) {
    const float a = s_ang[t], c = cosf(a), s = sinf(a);
    float ox, oy, oz;
    qrot_inv_y(c, s, vx_prev[nmine & 3], 0.f, vz_prev[nmine & 3], ox, oy, oz);
    s_x[t] = ox;
    s_z[t] = oz;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // r_pos = cumsum(r_pos, dim=-2)   (:382)
    float ax = 0.f, az = 0.f;
    for (int t = 0; t < T; ++t) {
      ax += s_x[t];
      az += s_z[t];
      s_x[t] = ax;
      s_z[t] = az;
    }
  }
  __syncthreads();
  float* ob = out + (size_t)b * J * 3 * T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float a = s_ang[t], c = cosf(a), s = sinf(a);
    const float px = s_x[t], pz = s_z[t];
    ob[(size_t)0 * T + t] = px;            // joint 0 = root: (x, data[..., 3], z)   (:384, :449)
    ob[(size_t)1 * T + t] = feat(3, t);
    ob[(size_t)2 * T + t] = pz;
    for (int j = 1; j < J; ++j) {          // positions = qrot(qinv(q), ric) + root xz   (:439-446)
      const int f = 4 + 3 * (j - 1);
      float ox, oy, oz;
      qrot_inv_y(c, s, feat(f, t), feat(f + 1, t), feat(f + 2, t), ox, oy, oz);
      ob[((size_t)j * 3 + 0) * T + t] = ox + px;
      ob[((size_t)j * 3 + 1) * T + t] = oy;
      ob[((size_t)j * 3 + 2) * T + t] = oz + pz;
    }
  }
}

}  // namespace mdm
