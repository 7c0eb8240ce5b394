// pi constants with the reference's digits (reference: src/constants.h:4-6)
#pragma once
#define PI_f     3.1415926535f
#define PI_X2_f  6.2831853072f
#define PI_D2_f  1.5707963267f
