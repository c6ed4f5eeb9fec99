export TMPDIR=/tmp
for e in "A=1" "AISGPU_FRONT_CUS=200" "AISGPU_FRONT_CUS=176" "AISGPU_FRONT_CUS=152" "AISGPU_FRONT_CUS=128" "AISGPU_FRONT_CUS=176 AISGPU_K4=lane" "AISGPU_FRONT_CUS=152 AISGPU_K4=lane" "AISGPU_FRONT_CUS=128 AISGPU_K4=lane" "AISGPU_FRONT_CUS=176 AISGPU_BACK_ALL=1"; do echo "$e"; env $e python bench.py --no-cpu-baseline | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['roofline'])"; done
