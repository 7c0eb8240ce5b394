// Sorting helpers for the KD-tree median (reference: src/utility.h, src/utility.cpp:5-41).
#pragma once
void fquicksort(float arr[], int left, int right);
float fmedian(float arr[], int size);
