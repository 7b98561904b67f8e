for abl in 0 4 1 5 2 6 7; do echo "ABL=$abl"; FX_C3S2_ABL=$abl python scripts/dev/conv_layer_bench.py 16,160,160,128,128,3,2 16,80,80,256,256,3,2 16,40,40,512,512,3,2 2>&1 | grep -v amdgpu; done
echo "igemm"; FX_C3S2_KPLANE=0 python scripts/dev/conv_layer_bench.py 16,160,160,128,128,3,2 16,80,80,256,256,3,2 16,40,40,512,512,3,2 2>&1 | grep -v amdgpu
echo "stride-1 reference points"; python scripts/dev/conv_layer_bench.py 16,80,80,128,128,3,1 16,40,40,256,256,3,1 16,20,20,512,512,3,1 2>&1 | grep -v amdgpu
