// Camera set-up and interactive controls behind the reference's entry points (reference:
// src/camera.cpp:15-128: Camera::init, reset, refresh, move, rotateV/H, the update* / toggle* setters).
//
// Written around two small pieces the reference spells out inline: the pixel grid of the projection
// (shared by init() and the headless overrideResolution()) and the turn of the three initial axes by the
// two view angles (one table-driven loop). The ARITHMETIC - which value is normalised when, which
// quotient is formed in float, the order of the two rotations - is the reference's, operation for
// operation: tests/test_oracle_pins.py compares every field bit for bit with the reference's own
// camera.cpp compiled unchanged (oracle/Makefile, target ref).
#include "camera.h"
#include "config.h"
#include "constants.h"
#include "glmlite.h"
#include "jsonResolve.h"

#include <cstdio>

namespace {

// Pixel grid for a screen of height `screen_height` (world units) at `w` x `h` pixels: square pixels of
// side screen_height / h; the half extents follow from the height (reference: src/camera.cpp:33-37).
template <typename Projection>
void layPixelGrid(Projection& p, int w, int h, float screen_height) {
	p.width = w;
	p.height = h;
	p.num_pixels = w * h;
	p.pixel_size = make_float2(screen_height / (float)h);
	p.screen_halfsize.y = screen_height / 2.0f;
	p.screen_halfsize.x = p.screen_halfsize.y * (float)w / (float)h;
}

// v turned by `rad` about `axis`, through a glm-style mat4 built from identity (reference:
// src/camera.cpp:59-66). A vector (anti)parallel to the axis is returned untouched: the reference's
// guard, which keeps the exact input bits instead of a rotation's rounding.
float3 turnAbout(float3 v, float rad, float3 axis) {
	if (fabsf(dot(v, axis)) == 1.0f) return v;
	const glmlite::mat4 r = glmlite::rotate(glmlite::mat4(), rad, glmlite::vec3{axis.x, axis.y, axis.z});
	const glmlite::vec4 t = glmlite::mul(r, glmlite::vec4{v.x, v.y, v.z, 1.0f});
	return make_float3(t.x, t.y, t.z);
}

// degrees -> radians with the reference's mixed precision: the quotient in double, the product in float
float radiansOf(float degrees) { return (float)(degrees / 180.0) * PI_f; }

} // namespace

void Camera::init(const Json::Value& jcam) {
	init_dir = normalize(resolveFloat3(jcam["direction"]));
	init_up = normalize(resolveFloat3(jcam["up"]));
	init_right = normalize(cross(init_dir, init_up));
	init_pos = resolveFloat3(jcam["position"]);

	layPixelGrid(projection, jcam["pixelWidth"].asInt(), jcam["pixelHeight"].asInt(), resolveFloat(jcam["screenHeight"]));
	projection.screen_dist = resolveFloat(jcam["screenDist"]);
	projection.aa_jitter = 2.0f;
	projection.gamma_corr = 0.5f;

	// the reference starts in preview mode with textures off (src/camera.cpp:41-45); a headless renderer
	// switches to path tracing through SceneState, as the GUI's "normal" radio button does
	preview_mode = true;
	aabb_mode = false;
	texture_enabled = false;
	max_ray_bounces = MAX_NUM_RAY_BOUNCES;

	reset();
	changed = false;
}

void Camera::overrideResolution(int width, int height) {
	if (width <= 0 || height <= 0) return;
	// screen_halfsize.y was formed as screenHeight / 2.0f, so doubling it gives screenHeight back exactly
	layPixelGrid(projection, width, height, projection.screen_halfsize.y * 2.0f);
	changed = true;
}

void Camera::reset() {
	position = init_pos;
	direction = init_dir;
	up = init_up;
	right = init_right;
	h_ang = v_ang = 0.0f;
	changed = true;
}

// Every axis of the view is its INITIAL axis turned first by the vertical angle about the initial right
// vector, then by the horizontal angle about the initial up vector, normalised after each turn
// (reference: src/camera.cpp:72-80).
void Camera::refresh() {
	const float rv = radiansOf(v_ang), rh = radiansOf(h_ang);
	const float3 from[3] = {init_dir, init_up, init_right};
	float3* const to[3] = {&direction, &up, &right};
	for (int k = 0; k < 3; ++k) {
		const float3 tilted = normalize(turnAbout(from[k], rv, init_right));
		*to[k] = normalize(turnAbout(tilted, rh, init_up));
	}
	changed = true;
}

void Camera::move(float3 diff) {
	position += diff;
	changed = true;
}

void Camera::rotateV(float degrees) {
	v_ang += degrees;
	refresh();
}

void Camera::rotateH(float degrees) {
	h_ang += degrees;
	refresh();
}

// The three bounded setters (reference: src/camera.cpp:98-117): a step that would leave the allowed range is
// ignored; gamma does not restart the accumulation (it only changes the display), the other two do.
void Camera::updateMaxRayBnc(int delta) {
	const int next = max_ray_bounces + delta;
	if (next < 0) return;
	max_ray_bounces = next;
	changed = true;
}

void Camera::updateAajitter(float delta) {
	const float next = projection.aa_jitter + delta;
	if (next < 0.0) return;
	projection.aa_jitter = next;
	changed = true;
}

void Camera::updateGamma(float delta) {
	const float next = projection.gamma_corr + delta;
	if (next <= 0.0) return;
	projection.gamma_corr = next;
}

void Camera::togglePrevMode() {
	preview_mode = !preview_mode;
	changed = true;
}

void Camera::toggleTextures() {
	texture_enabled = !texture_enabled;
	changed = true;
}

void Camera::printInfo() {
	const struct { const char* name; float3 v; } rows[] = {{"position: ", position}, {"direction:", direction}, {"up:       ", up}, {"right:    ", right}};
	printf("\nCamera Info -------------------\n");
	for (const auto& r : rows) printf("> %s [%.3f,%.3f,%.3f]\n", r.name, r.v.x, r.v.y, r.v.z);
	printf("> v_ang:            %.3f\n> h_ang:            %.3f\n> max_ray_bounces:  %d\n> aa ray jitter:    %.3f\n> image gamma:      %.3f\n",
	       v_ang, h_ang, max_ray_bounces, projection.aa_jitter, projection.gamma_corr);
}
