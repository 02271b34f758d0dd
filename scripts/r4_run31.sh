cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
for w in wing1m wing4m; do
run 600 gpurun_out/r4z_ab2_$w.txt python scripts/r4_ab.py $w "super_panel_ahead=0,diag_ahead=0,super_tiles=0" "super_panel_ahead=1,diag_ahead=0,super_tiles=0" "super_panel_ahead=0,diag_ahead=1,super_tiles=0" "super_panel_ahead=0,diag_ahead=0,super_tiles=1"
cat gpurun_out/r4z_ab2_$w.txt
done
