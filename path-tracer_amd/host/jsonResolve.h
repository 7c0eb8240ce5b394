// JSON → float/vector/material-type helpers (reference: src/jsonResolve.h:6-10).
#pragma once
#include "world.h"
#include "glmlite.h"

float3 resolveFloat3(const Json::Value&);
glmlite::vec3 resolveGlmVec3(const Json::Value&);
float resolveFloat(const Json::Value&);
MaterialType resolveMatType(const Json::Value&);
MaterialType resolveMatType(const std::string&);
