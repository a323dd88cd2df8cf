# Does the command-line leg run slower inside bench.py than alone?  (small parent, then the default parent)
show() { grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['e2e']['cli_from_fasta']
print(sys.argv[1], {k: c[k] for k in ('wall_s', 'first_rows_after_s', 'seconds_typing', 'exit_s')}, c['phases_s'], 'small:', {k: c['about_1000_files'][k] for k in ('wall_s', 'first_rows_after_s', 'exit_s')}, c['about_1000_files']['phases_s'])
" "$1"; }
python bench.py --assemblies 1000 --steps 1 --warmup 0 --no-cpu-baseline --e2e-steps 1 2>/dev/null | show small-parent
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-steps 1 2>/dev/null | show default-parent
