# bash tools/big_ab.sh   (GPU box) - tools/big_scene_check.py 8 16 with the product library and every experiment build lying in the package directory
for lib in $PWD/rsoderh-raytracing_amd/librsrt.so $PWD/rsoderh-raytracing_amd/librsrt_exp_*.so; do echo "== $lib"; RSRT_LIB=$lib timeout -k 10 300 python tools/big_scene_check.py 8 16 | grep "bit-exact\|TRAVERSAL=4"; done
