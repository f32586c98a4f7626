// tracer_amd/csrc/host/Camera.cpp -- see Camera.h. Arithmetic follows reference
// src/Camera.cpp line by cited line so that positions and bases match its host build bit for bit.
#include "Camera.h"

#pragma clang fp contract(off)

using namespace trc;

Camera::Camera()
  : m_lensRadius(1.0f), m_focalDist(5.0f),
    m_currXZRads(0.0f), m_currYRads(0.0f),
    m_position(splat3(0.0f)), m_upDir(mk3(0.0f, 1.0f, 0.0f)), m_lookDir(mk3(0.0f, 0.0f, 1.0f)),
    m_rightDir(cross(mk3(0.0f, 0.0f, 1.0f), mk3(0.0f, 1.0f, 0.0f))), m_lookAt(splat3(0.0f)),
    m_nearPlane(0.1f), m_farPlane(1000.0f), m_fov(45.0f), m_aspect(1.0f),
    m_rotationAmount(0.005f), m_panAmount(0.04f), m_zoomAmount(0.05f) {}   // Camera.cpp:3-20

Camera::~Camera() {}

void Camera::rotateDegrees(float x, float y) { rotate(radians(x), radians(y)); }            // :26-29
void Camera::setAspect(unsigned w, unsigned h) { m_aspect = (float)w / h; }                  // :31-34
void Camera::setFarNearPlanes(float n, float f) { m_nearPlane = n; m_farPlane = f; }         // :35-39
void Camera::setFOV(float fov) { m_fov = fov; }                                              // :40-43

void Camera::orbitFrame(f3& dir, f3& pitchAxis) const {
  const float cosa = cosf(m_currXZRads);
  const float sina = sinf(m_currXZRads);
  const f3 flat = mk3(sina, 0.0f, cosa);
  pitchAxis = mk3(flat.z, flat.y, -flat.x);
  const mat4 pitch = trc::rotate(identity4(), degrees(m_currYRads), pitchAxis);
  const f4 d = mul(pitch, mk4(flat.x, flat.y, flat.z, 0.0f));
  dir = mk3(d.x, d.y, d.z);
}

// ---- orbit camera -------------------------------------------------------------------------
ThirdPersonCamera::ThirdPersonCamera() : Camera(), m_radius(0.0f) {}

void ThirdPersonCamera::rotate(float x, float y) { m_currXZRads += x; m_currYRads += y; }    // :79-83
void ThirdPersonCamera::pan(float x, float y) {                                              // :85-89
  m_lookAt = m_lookAt + x * normalize(cross(m_lookDir, m_upDir));
  m_lookAt = m_lookAt + y * m_upDir;
}
void ThirdPersonCamera::zoom(float distance) { m_radius -= distance; }                       // :91-94
void ThirdPersonCamera::update() {                                                           // :96-112
  f3 dir, axis;
  orbitFrame(dir, axis);
  m_position = dir * m_radius + m_lookAt;
  m_upDir = normalize(cross(dir, axis));
  m_lookDir = normalize(m_lookAt - m_position);
  m_rightDir = cross(m_lookDir, m_upDir);
}

// ---- fly camera ---------------------------------------------------------------------------
FirstPersonCamera::FirstPersonCamera() : Camera() {}

void FirstPersonCamera::rotate(float x, float y) { m_currXZRads += x; m_currYRads -= y; }    // :126-130
void FirstPersonCamera::pan(float x, float y) {                                              // :132-136
  m_position = m_position + x * normalize(cross(m_lookDir, m_upDir));
  m_position = m_position + y * m_upDir;
}
void FirstPersonCamera::zoom(float distance) { m_position = m_position + distance * m_lookDir; } // :138-141
void FirstPersonCamera::update() {                                                           // :143-157
  f3 dir, axis;
  orbitFrame(dir, axis);
  m_lookDir = normalize(dir);
  m_upDir = normalize(cross(dir, axis));
  m_rightDir = cross(m_lookDir, m_upDir);
}
