cd $GRAFT_REPO_ROOT
i=0
for s in "-" "f8:fic=1;b8:epi=1;b6:epi=1;f6:fic=1;b7:epi=1;b5:epi=1;f2:fic=0;f5:fic=0" "b8:epi=2;b6:epi=2;b7:epi=2;b5:epi=2;b7:fic=0" "b8:epi=0;b6:epi=0;b7:epi=0;b5:epi=0;f4:fic=1;b2:fic=1;b4:fic=1"; do
  if [ "$s" = "-" ]; then unset ALQ_G4_TUNE; else export ALQ_G4_TUNE="$s"; fi
  ALQ_DUMP_ARGS=1 python bench.py --pool 4000 --steps 1 --warmup 0 --no-cpu-baseline --netb-pool 0 > /dev/null 2> gpurun_out/tunedump_$i.err
  grep -c G4ARGS gpurun_out/tunedump_$i.err
  i=$((i+1))
done
ALQ_ALT16=1 ALQ_DUMP_ARGS=1 python bench.py --pool 4000 --steps 1 --warmup 0 --no-cpu-baseline --netb-pool 0 > /dev/null 2> gpurun_out/tunedump_alt.err
