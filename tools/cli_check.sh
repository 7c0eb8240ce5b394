# pt_render end to end: autosave naming, checkpoint, resume == uninterrupted (PFM bytes identical)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/cli; rm -rf $O; mkdir -p $O
R=path-tracer_amd/pt_render
$R --scene scenes/scene_geoms.json --width 320 --height 240 --spp 24 --pfm $O/whole.pfm --out $O/whole.png > $O/whole.log
$R --scene scenes/scene_geoms.json --width 320 --height 240 --spp 10 --checkpoint $O/ck.ptxacc --autosave 5 --autosave-dir $O --name geoms > $O/part.log
$R --scene scenes/scene_geoms.json --width 320 --height 240 --spp 24 --resume $O/ck.ptxacc --pfm $O/resumed.pfm > $O/resumed.log
cmp $O/whole.pfm $O/resumed.pfm && echo "resume bit-identical"
ls $O | grep spp.png
$R --scene scenes/scene_teapot.json --width 160 --height 120 --spp 6 --iterate --autosave 4 --autosave-dir $O --name teapot_it | grep -c saved
tail -2 $O/whole.log
