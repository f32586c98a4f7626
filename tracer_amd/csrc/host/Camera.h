// tracer_amd/csrc/host/Camera.h -- the reference's camera interface (src/Camera.h:9-80),
// GLM-free: Camera (abstract), ThirdPersonCamera (orbit), FirstPersonCamera. Same public
// names and meanings: angles in radians for rotate(), degrees for rotateDegrees()/setFOV();
// m_fov is the HALF-angle the renderer takes the tangent of (Renderer.cpp:168).
#pragma once
#include "linalg.h"

class Camera {
public:
  Camera();
  virtual ~Camera();

  void rotateDegrees(float x, float y);
  virtual void rotate(float x, float y) = 0;
  virtual void pan(float x, float y) = 0;
  virtual void zoom(float delta) = 0;
  virtual void update() = 0;

  void setAspect(unsigned screenWidth, unsigned screenHeight);
  void setFOV(float fov);
  void setFarNearPlanes(float nearPlane, float farPlane);

  trc::f3 getPosition() const { return m_position; }
  trc::f3 getLookAt() const { return m_lookAt; }
  trc::f3 getUp() const { return m_upDir; }
  float getAspect() const { return m_aspect; }
  float getFOV() const { return m_fov; }

  float m_lensRadius;   // thin-lens aperture radius, default 1 (Camera.cpp:17)
  float m_focalDist;    // distance to the plane in focus, default 5 (Camera.cpp:18)

protected:
  // orientation shared by both cameras: the unit vector at (m_currXZRads, m_currYRads) and the
  // horizontal axis it was pitched about (Camera.cpp:98-105, :145-152)
  void orbitFrame(trc::f3& dir, trc::f3& pitchAxis) const;

  float m_currXZRads, m_currYRads;
  trc::f3 m_position, m_upDir, m_lookDir, m_rightDir, m_lookAt;
  float m_nearPlane, m_farPlane;
  float m_fov;      // degrees
  float m_aspect;
  float m_rotationAmount, m_panAmount, m_zoomAmount;
};

struct ThirdPersonCamera : public Camera {
  ThirdPersonCamera();
  void rotate(float x, float y) override;
  void pan(float x, float y) override;
  void zoom(float distance) override;
  void update() override;
protected:
  float m_radius;
};

struct FirstPersonCamera : public Camera {
  FirstPersonCamera();
  void rotate(float x, float y) override;
  void pan(float x, float y) override;
  void zoom(float distance) override;
  void update() override;
};
