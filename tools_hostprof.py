import cProfile, pstats, sys, os
sys.argv = ['bench.py', '--steps', '1', '--warmup', '1', '--cpu-procs', '0', '--no-profile']
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(60)
