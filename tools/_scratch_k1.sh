for q in 3 2 1; do
export GPU_MAX_HW_QUEUES=$q
timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/q$q.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('queues=%r' % '$q', d['ms_per_step'], d['ms_per_step_hip_graph'], d['ms_per_step_with_h2d'], d['ms_per_step_all_fps_rounds'])"
echo "rc=$? q=$q"; tail -3 gpurun_out/q$q.err | cut -c1-300
done
