cd /root/repo
echo "=== S=1 as compiled (packed fp32), 300 x 8 launches beside the unplanned QuantLinear kernels"
IVIT_LIB=$PWD/build/lnprobe/libivit_s1v0.so MIXED_ONLY=1 SWIN_ONLY=1 MIX_FILTERS="ivit_linear_i8_requant" timeout 900 python tools/op_stress.py 300 8 2>&1 | grep -v amdgpu.ids | grep -v "^    " | tail -4
echo "=== S=1 built without packed fp32, 2500 x 8 = 20000 launches"
IVIT_LIB=$PWD/build/lnprobe/libivit_s1v4.so MIXED_ONLY=1 SWIN_ONLY=1 MIX_FILTERS="ivit_linear_i8_requant" timeout 1500 python tools/op_stress.py 2500 8 2>&1 | grep -v amdgpu.ids | grep -v "^    " | tail -4
echo "=== production library (packed fp32 in the S=4 forms), victim C=768 M=12608, 600 x 8"
LN_BIG=1 MIX_VICTIM="layernorm_requant C=768 M=12608" MIXED_ONLY=1 SWIN_ONLY=1 MIX_FILTERS="ivit_linear_i8_requant" timeout 900 python tools/op_stress.py 600 8 2>&1 | grep -v amdgpu.ids | grep -v "^    " | tail -4
echo "=== production library, victim C=1024 M=6304, 600 x 8"
LN_BIG=1 MIX_VICTIM="layernorm_requant C=1024 M=6304" MIXED_ONLY=1 SWIN_ONLY=1 MIX_FILTERS="ivit_linear_i8_requant" timeout 900 python tools/op_stress.py 600 8 2>&1 | grep -v amdgpu.ids | grep -v "^    " | tail -4
