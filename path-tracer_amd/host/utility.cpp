// Median used by the KD-tree split (reference: src/utility.cpp:5-41).
// The split plane is a float compared against triangle centroids, so the value
// must be exactly the reference's: sort ascending, odd -> middle element, even
// -> (a[n/2-1] + a[n/2]) / 2.0f. Any correct ascending sort yields the same
// sorted array (floats compare by value; -0/+0 ties cannot change a sum by more
// than its sign of zero, and NaN never appears in finite meshes), so the sort
// itself is a plain Hoare-partition quicksort like the reference's.
#include "utility.h"

void fquicksort(float arr[], int left, int right) {
	while (left < right) {
		int i = left, j = right;
		const float pivot = arr[left + (right - left) / 2];
		while (i <= j) {
			while (arr[i] < pivot) ++i;
			while (arr[j] > pivot) --j;
			if (i <= j) {
				float t = arr[i]; arr[i] = arr[j]; arr[j] = t;
				++i; --j;
			}
		}
		// recurse into the smaller part, loop on the larger (bounded stack on 1M+ inputs)
		if (j - left < right - i) {
			if (left < j) fquicksort(arr, left, j);
			left = i;
		} else {
			if (i < right) fquicksort(arr, i, right);
			right = j;
		}
	}
}

float fmedian(float arr[], int size) {
	if (size < 1) return 0.0f;
	fquicksort(arr, 0, size - 1);
	int halfs = size / 2;
	if (size % 2 == 0) return (arr[halfs - 1] + arr[halfs]) / 2.0f;
	return arr[halfs];
}
