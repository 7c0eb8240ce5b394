// Headless driver: the reference's render thread (reference: qt/cudarunthread.cpp:14-56) and
// scene loader thread (qt/sceneloaderthread.cpp:11-23) as a command-line program.
//   pt_render --scene scenes/scene_geoms.json [--width W --height H] [--spp N] [--bounces B]
//             [--textures 0|1] [--seed S] [--device D] [--batch K] [--out image.png] [--pfm image.pfm]
//             [--assets DIR] [--car-tris N] [--iterate]
//             [--autosave N --autosave-dir DIR --name NAME]   PNG every N spp, reference file naming
//             [--checkpoint FILE [--checkpoint-every N]] [--resume FILE]   float accumulation checkpoints
// --iterate drives runCUDA() once per sample exactly like the reference loop (tonemap every
// iteration); the default renders the samples in batches of K per launch.
#include "kernel.h"
#include "scenestate.h"
#include "world_load.h"
#include "image_io.h"
#include "checkpoint.h"
#include "hostscene.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char** argv) {
	std::string scene_path, out_png, out_pfm, assets;
	int width = 0, height = 0, spp = 16, bounces = -1, textures = -1, batch = 128, car_tris = 0;
	unsigned long long seed = 0;
	bool iterate = false;
	std::string autosave_dir = ".", name = "render", ck_path, resume_path;
	int autosave = 0, ck_every = 0;
	for (int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		auto next = [&](const char* what) -> const char* {
			if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", what); exit(2); }
			return argv[++i];
		};
		if (a == "--scene") scene_path = next("--scene");
		else if (a == "--width") width = atoi(next("--width"));
		else if (a == "--height") height = atoi(next("--height"));
		else if (a == "--spp") spp = atoi(next("--spp"));
		else if (a == "--bounces") bounces = atoi(next("--bounces"));
		else if (a == "--textures") textures = atoi(next("--textures"));
		else if (a == "--seed") seed = strtoull(next("--seed"), NULL, 10);
		else if (a == "--device") hip_device_id = atoi(next("--device"));
		else if (a == "--batch") batch = atoi(next("--batch"));
		else if (a == "--out") out_png = next("--out");
		else if (a == "--pfm") out_pfm = next("--pfm");
		else if (a == "--assets") assets = next("--assets");
		else if (a == "--car-tris") car_tris = atoi(next("--car-tris"));
		else if (a == "--iterate") iterate = true;
		else if (a == "--autosave") autosave = atoi(next("--autosave"));
		else if (a == "--autosave-dir") autosave_dir = next("--autosave-dir");
		else if (a == "--name") name = next("--name");
		else if (a == "--checkpoint") ck_path = next("--checkpoint");
		else if (a == "--checkpoint-every") ck_every = atoi(next("--checkpoint-every"));
		else if (a == "--resume") resume_path = next("--resume");
		else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
	}
	if (scene_path.empty()) { fprintf(stderr, "usage: pt_render --scene file.json [options]\n"); return 2; }
	printProgramConfig();
	if (detectHIPdevice() < 0) return 1;
	if (!assets.empty()) setAssetRoot(assets);
	if (car_tris > 0) setProceduralCarTriangles(car_tris);
	kernelSetSeed(seed);

	Json::Value jscene;
	std::string err;
	if (!parseJsonScene(scene_path, jscene, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
	DialogLogger logger(stdout);
	if (!sceneState.load(jscene, &logger)) return 1;
	if (width > 0 && height > 0) sceneState.overrideResolution(width, height);
	sceneState.togglePrevMode(false);
	sceneState.toggleAABBmode(false);
	if (textures >= 0) sceneState.toggleTextures(textures != 0);
	if (bounces >= 0) sceneState.updateMaxRayBnc(bounces);

	Scene scene = sceneState.clone();
	sceneState.cameraChanged(false, NULL);
	scene.camera.changed = false;
	const int W = scene.camera.projection.width, H = scene.camera.projection.height;
	kernelInit(scene);
	if (checkCudaError("kernelInit")) return 1;
	// the display buffer: page-locked (kernelAllocPBO), as the reference's mapped pixel buffer is device-visible
	struct Pbo {
		uchar4* p; size_t n;
		uchar4* data() { return p; }
		const uchar4& operator[](size_t i) const { return p[i]; }
	} pbo = {kernelAllocPBO((size_t)W * H), (size_t)W * H};
	if (!pbo.p) { fprintf(stderr, "display buffer: %s\n", ptx_last_error()); return 1; }
	pbo_dptr = pbo.data();

	PtxCtx* ctx = kernelContext();
	// on-screen orientation of the reference: buffer mirrored horizontally (qt/previewglwidget.cpp:111-116)
	auto save_png = [&](const std::string& file) {
		std::vector<unsigned char> rgb((size_t)W * H * 3);
		for (int y = 0; y < H; ++y)
			for (int x = 0; x < W; ++x) {
				const uchar4& p = pbo[(size_t)y * W + (W - 1 - x)];
				unsigned char* o = &rgb[((size_t)y * W + x) * 3];
				o[0] = p.x; o[1] = p.y; o[2] = p.z;
			}
		if (!writePNG(file, W, H, 3, rgb.data(), err)) fprintf(stderr, "%s\n", err.c_str());
		else printf("saved %s\n", file.c_str());
	};
	auto save_checkpoint = [&](int done) {
		AccumCheckpoint ck;
		ck.width = W; ck.height = H; ck.samples_done = done; ck.seed = seed;
		ck.scene_hash = renderContentHash(hostSceneFlat().desc, scene.camera);
		ck.max_ray_bounces = scene.camera.max_ray_bounces;
		ck.texture_enabled = scene.camera.texture_enabled ? 1 : 0;
		ck.rgba.resize((size_t)W * H * 4);
		if (ptx_accum_download(ctx, ck.rgba.data()) != PTX_OK || !saveCheckpoint(ck_path, ck, err))
			fprintf(stderr, "checkpoint failed: %s %s\n", err.c_str(), ptx_last_error());
	};
	int start = 0;
	ptx_accum_clear(ctx);
	if (!resume_path.empty()) {
		AccumCheckpoint ck;
		if (!loadCheckpoint(resume_path, ck, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
		if (ck.width != W || ck.height != H || ck.seed != seed) {
			fprintf(stderr, "checkpoint %s is %dx%d seed %llu, this run is %dx%d seed %llu\n", resume_path.c_str(), ck.width, ck.height,
			        (unsigned long long)ck.seed, W, H, seed);
			return 1;
		}
		const std::string why = checkpointMismatch(ck, renderContentHash(hostSceneFlat().desc, scene.camera), scene.camera.max_ray_bounces,
		                                           scene.camera.texture_enabled ? 1 : 0);
		if (!why.empty()) { fprintf(stderr, "checkpoint %s does not continue this render: %s\n", resume_path.c_str(), why.c_str()); return 1; }
		if (ck.samples_done > spp) {
			// the buffer already holds more samples than asked for: dividing it by --spp would brighten the image and
			// saving it as "spp samples" would make the next resume count samples twice
			fprintf(stderr, "checkpoint %s holds %d samples, more than --spp %d: nothing to render (ask for at least %d)\n",
			        resume_path.c_str(), ck.samples_done, spp, ck.samples_done);
			return 1;
		}
		if (ptx_accum_upload(ctx, ck.rgba.data()) != PTX_OK) { fprintf(stderr, "%s\n", ptx_last_error()); return 1; }
		start = ck.samples_done;
		printf("resumed %s at %d spp\n", resume_path.c_str(), start);
	}
	auto t0 = std::chrono::steady_clock::now();
	if (iterate) {
		for (int i = start + 1; i <= spp; ++i) {
			if (i == 1) runCUDA(pbo_dptr, scene, i);   // iteration 1 clears the buffer, as in the reference
			else {
				if (renderOneSample(scene, i - 1) != PTX_OK) { fprintf(stderr, "%s\n", ptx_last_error()); return 1; }
				ptx_tonemap(ctx, i, scene.camera.projection.gamma_corr, (unsigned char*)pbo.data());
			}
			if (checkCudaError("runCUDA")) return 1;
			if (autosaveDue(i, autosave)) save_png(autosaveFileName(autosave_dir, name, i));
			if (!ck_path.empty() && ck_every > 0 && i % ck_every == 0) save_checkpoint(i);
		}
	} else {
		int s = start;
		while (s < spp) {
			int n = spp - s < batch ? spp - s : batch;
			// stop a batch at the next autosave / checkpoint boundary
			if (autosave > 0 && (s / autosave + 1) * autosave - s < n) n = (s / autosave + 1) * autosave - s;
			if (ck_every > 0 && (s / ck_every + 1) * ck_every - s < n) n = (s / ck_every + 1) * ck_every - s;
			if (renderSamples(scene, s, n, false) != PTX_OK) { fprintf(stderr, "%s\n", ptx_last_error()); return 1; }
			s += n;
			if (autosave > 0 && s % autosave == 0) {
				ptx_tonemap(ctx, s, scene.camera.projection.gamma_corr, (unsigned char*)pbo.data());
				save_png(autosaveFileName(autosave_dir, name, s));
			}
			if (!ck_path.empty() && ck_every > 0 && s % ck_every == 0) save_checkpoint(s);
		}
		ptx_render_sync(ctx);
		ptx_tonemap(ctx, spp, scene.camera.projection.gamma_corr, (unsigned char*)pbo.data());
	}
	if (!ck_path.empty()) save_checkpoint(spp);
	double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	PtxStats st;
	ptx_stats_get(ctx, &st);
	printf("rendered %dx%d, %d spp in %.3f s: %.1f Mrays/s, %.2f Msamples/s\n", W, H, spp, sec, st.rays / sec * 1e-6,
	       (double)W * H * spp / sec * 1e-6);

	if (!out_png.empty()) save_png(out_png);
	if (!out_pfm.empty()) {
		std::vector<float> acc((size_t)W * H * 4), rgbf((size_t)W * H * 3);
		ptx_accum_download(ctx, acc.data());
		for (size_t i = 0; i < (size_t)W * H; ++i)
			for (int k = 0; k < 3; ++k) rgbf[3 * i + k] = acc[4 * i + k] / (float)spp;
		if (!writePFM(out_pfm, W, H, rgbf.data(), err)) fprintf(stderr, "%s\n", err.c_str());
	}
	kernelCleanUp();
	kernelFreePBO(pbo.p);
	sceneState.clean();
	kernelShutdown();
	return 0;
}
