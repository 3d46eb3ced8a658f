run() { python bench.py --model $2 --no-cpu-baseline --measure-traffic 0 --profile-steps 0 --box-probe 0 --min-seconds 1 $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 $3:', d['ms_per_step'], d['value'], d['config'].get('streams_per_gpu'), d.get('bit_exact_vs_reference_golden'), d.get('all_images_equal_unsliced_forward'))"; }
for i in 1 2 3; do
  for m in swin_tiny; do run tree $m; IVIT_LIB=build/ab/noshare.so run noshare $m; done
done
for i in 1 2; do
  run tree deit_small "--streams 2 --graph 1"; IVIT_LIB=build/ab/noshare.so run noshare deit_small "--streams 2 --graph 1"
  run tree deit_small; IVIT_LIB=build/ab/noshare.so run noshare deit_small
  run tree deit_base; IVIT_LIB=build/ab/noshare.so run noshare deit_base
  run tree vit_base_384; IVIT_LIB=build/ab/noshare.so run noshare vit_base_384
done
