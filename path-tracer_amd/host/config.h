// Compile-time parameters of the path tracer (reference: src/config.h:4-21, same values).
#pragma once

// cam properties
#define CAM_ROTATE_ANGLE_DELTA   5.0f  // degrees
#define CAM_MOVE_DISTANCE_DELTA  0.2f  // units

// path tracing parameters
#define MAX_NUM_RAY_BOUNCES      5     // maximum number of traced path segments
#define SCENE_REFRACTION_INDEX   1.0f  // refraction index of empty space

// KD-tree: a node is split only while its "left" half holds more than this many triangles
#define MAX_KD_LEAF_TRGS         16
