O=gpurun_out/r05_ab; mkdir -p $O; B=tools/dev/_build
timeout 200 python tools/dev/ab_zero.py > $O/strip_abl_base.txt 2>&1
for v in noload nostage noepi nobarrier; do GARMENTNETS_HIP_LIB=$B/lib_st_$v.so timeout 200 python tools/dev/ab_zero.py > $O/strip_abl_$v.txt 2>&1; done
for v in base noload nostage noepi nobarrier; do echo == $v; grep TF $O/strip_abl_$v.txt | cut -c1-110; done
