#include "checkpoint.h"

#include <cstdio>
#include <cstring>
#include <ctime>

static const char CK_MAGIC[8] = {'P', 'T', 'X', 'A', 'C', 'C', '1', 0};

bool saveCheckpoint(const std::string& path, const AccumCheckpoint& ck, std::string& err) {
	if (ck.width <= 0 || ck.height <= 0 || ck.samples_done < 0 || ck.rgba.size() != (size_t)ck.width * ck.height * 4) {
		err = "saveCheckpoint: inconsistent checkpoint";
		return false;
	}
	const std::string tmp = path + ".tmp";
	FILE* f = fopen(tmp.c_str(), "wb");
	if (!f) { err = "saveCheckpoint: cannot open " + tmp; return false; }
	bool ok = fwrite(CK_MAGIC, 1, 8, f) == 8 && fwrite(&ck.width, 4, 1, f) == 1 && fwrite(&ck.height, 4, 1, f) == 1 &&
	          fwrite(&ck.samples_done, 4, 1, f) == 1 && fwrite(&ck.seed, 8, 1, f) == 1 &&
	          fwrite(ck.rgba.data(), sizeof(float), ck.rgba.size(), f) == ck.rgba.size();
	ok = (fclose(f) == 0) && ok;
	// written beside the target and renamed, so an interrupted save never clobbers the last good checkpoint
	if (ok && rename(tmp.c_str(), path.c_str()) != 0) ok = false;
	if (!ok) { remove(tmp.c_str()); err = "saveCheckpoint: write failed for " + path; }
	return ok;
}

bool loadCheckpoint(const std::string& path, AccumCheckpoint& ck, std::string& err) {
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) { err = "loadCheckpoint: cannot open " + path; return false; }
	char magic[8];
	bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, CK_MAGIC, 8) == 0;
	if (!ok) { fclose(f); err = "loadCheckpoint: " + path + " is not a PTXACC1 checkpoint"; return false; }
	ok = fread(&ck.width, 4, 1, f) == 1 && fread(&ck.height, 4, 1, f) == 1 && fread(&ck.samples_done, 4, 1, f) == 1 &&
	     fread(&ck.seed, 8, 1, f) == 1;
	if (!ok || ck.width <= 0 || ck.height <= 0 || ck.samples_done < 0 || (int64_t)ck.width * ck.height > ((int64_t)1 << 28)) {
		fclose(f);
		err = "loadCheckpoint: bad header in " + path;
		return false;
	}
	ck.rgba.resize((size_t)ck.width * ck.height * 4);
	ok = fread(ck.rgba.data(), sizeof(float), ck.rgba.size(), f) == ck.rgba.size();
	char extra;
	if (ok && fread(&extra, 1, 1, f) != 0) ok = false; // trailing bytes: not ours
	fclose(f);
	if (!ok) err = "loadCheckpoint: truncated or oversized payload in " + path;
	return ok;
}

bool autosaveDue(int iter, int every) { return iter > 0 && every > 0 && ((iter % every) == 0 || iter == 1); }

std::string autosaveFileName(const std::string& folder, const std::string& name, int iter, int day, int month, int year) {
	char buf[64];
	snprintf(buf, sizeof(buf), "_%02d-%02d-%04d_%dspp.png", day, month, year, iter);
	return folder + "/" + name + buf;
}

std::string autosaveFileName(const std::string& folder, const std::string& name, int iter) {
	time_t now = time(NULL);
	struct tm tmv;
	localtime_r(&now, &tmv);
	return autosaveFileName(folder, name, iter, tmv.tm_mday, tmv.tm_mon + 1, tmv.tm_year + 1900);
}
