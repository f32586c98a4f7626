// tracer_amd/csrc/shade.h -- sampling and material response for gfx950: reference
// src/Utils.inl and src/Material.inl in their exact operation order.
#pragma once
#include "scene.h"
#include "pmath.h"

#pragma clang fp contract(off)

namespace trc {

// Utils.inl:15-23: uniform direction on the sphere. Note (r1*2)*pi.
TRC_HD f3 rand_vector(float rand1, float rand2) {
  float phi = rand1 * 2.0f * kPi;
  float theta = pm_acosf(rand2 * 2.0f - 1.0f);
  float st, ct, sp, cp;
  pm_sincosf(theta, st, ct);
  pm_sincosf(phi, sp, cp);
  return mk3(st * cp, st * sp, ct);
}

// Utils.inl:25-30: triangular-distributed radius (not sqrt-uniform). Note (2*pi)*r1.
TRC_HD f2 rand_point_disk(float rand1, float rand2, float rand3) {
  float t = 2.0f * kPi * rand1;
  float u = rand2 + rand3;
  float r = u > 1.0f ? 2.0f - u : u;
  float s, c;
  pm_sincosf(t, s, c);
  f2 o; o.x = r * c; o.y = r * s;
  return o;
}

// Utils.inl:32-37
TRC_HD f3 rand_vector_hem(float rand1, float rand2, f3 nor) {
  f3 v = rand_vector(rand1, rand2);
  if (dot(v, nor) < 0) v = -v;
  return v;
}

// Material.inl:58-67: exact unpolarised Fresnel; 1 on total internal reflection
// `n` = n1 / n2, formed by the caller (it is 1/ior or ior: one of two per-object constants).
TRC_HD float reflectance(f3 inc, f3 nor, float n1, float n2, float n) {
  float cosI = -dot(nor, inc);
  float sinT2 = n * n * (1.0f - cosI * cosI);
  if (sinT2 > 1.0f) return 1.0f;
  float cosT = pm_sqrtf(1.0f - sinT2);
  float rOrth = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
  float rPar = (n2 * cosI - n1 * cosT) / (n2 * cosI + n1 * cosT);
  return (rOrth * rOrth + rPar * rPar) / 2.0f;
}

// Material.inl:69-106
// core/func_geometric.inl:311-317 (vecmath.h: refract) with the device's short square root
TRC_HD f3 refract_pm(f3 I, f3 N, float eta) {
  float dotValue = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - dotValue * dotValue);
  if (k < 0.0f) return splat3(0.0f);
  return eta * I - (eta * dotValue + pm_sqrtf(k)) * N;
}

// `inv_ior` = 1.0f / ior, divided once per object at upload: n1 / n2 of Material.inl:100, :104 is 1/ior from
// outside and ior / 1 = ior from inside.
TRC_HD f3 bounce_trans(float ior, float inv_ior, f3 rd, f3 nor, f3 randvec) {
  float n1, n2, eta; f3 nnor;
  if (dot(rd, nor) < 0.0f) { n1 = 1.0f; n2 = ior; eta = inv_ior; nnor = nor; }     // from outside
  else                     { n1 = ior; n2 = 1.0f; eta = ior; nnor = -nor; }        // from inside
  float refl = reflectance(rd, nnor, n1, n2, eta);
  if (randvec.x < refl) return reflect(rd, nnor);
  return refract_pm(rd, nnor, eta);
}
#if defined(__HIP_DEVICE_COMPILE__) && (defined(TRC_DUP_TRANS) || defined(TRC_DUP_DIFF))
// cost-measuring builds (tools/pmc_dup.sh): the named branch is evaluated twice, the second time on inputs the
// optimiser cannot see through, so that the difference in SQ_INSTS_VALU is that branch's dynamic cost
__device__ __forceinline__ f3 opaque3(f3 v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z)); return v; }
#endif
TRC_HD f3 bounce(int mat_type, float ior, float inv_ior, f3 rd, f3 nor, f3 randvec) {
  if (mat_type == REF_DIFF) {
    f3 r = rand_vector_hem(randvec.x, randvec.y, nor);
#if defined(__HIP_DEVICE_COMPILE__) && defined(TRC_DUP_DIFF)
    f3 r2 = rand_vector_hem(opaque3(randvec).x, opaque3(randvec).y, opaque3(nor));
    if (r2.x != r.x) r = r2;
#endif
    return r;
  } else if (mat_type == REF_MIRR) {
    return reflect(rd, nor);
  } else {
    f3 r = bounce_trans(ior, inv_ior, rd, nor, randvec);
#if defined(__HIP_DEVICE_COMPILE__) && defined(TRC_DUP_TRANS)
    f3 r2 = bounce_trans(ior, inv_ior, opaque3(rd), opaque3(nor), opaque3(randvec));
    if (r2.x != r.x) r = r2;
#endif
    return r;
  }
}

// raytrace.cu:21-27: clamp, int(c*255.0) in double with truncation, 0x00RRGGBB
TRC_HD uint32_t rgb_to_int(f3 c) {
  float r = gclamp(c.x, 0.0f, 1.0f), g = gclamp(c.y, 0.0f, 1.0f), b = gclamp(c.z, 0.0f, 1.0f);
  return (uint32_t)(((int)(r * 255.0) << 16) | ((int)(g * 255.0) << 8) | (int)(b * 255.0));
}

// raytrace.cu:173: display transform of the accumulated film
TRC_HD uint32_t tonemap(f3 film, float iters) {
  const float g = 1.0f / 2.2f;
  f3 v = mk3(film.x / iters, film.y / iters, film.z / iters);
  return rgb_to_int(mk3(pm_powf(v.x, g), pm_powf(v.y, g), pm_powf(v.z, g)));
}

} // namespace trc
