# timing experiment (WRONG RESULTS on purpose): k_dec_b4 reduced to its ConvT3 contraction: no gather, no tap contraction, no pre-sums / H writes (barriers, staging and fragments kept)
import runpy, os
_d = os.path.dirname(os.path.abspath(__file__))
PATCH = {'decoder.hip': sum((runpy.run_path(os.path.join(_d, f))['PATCH']['decoder.hip'] for f in ('db_nogather.py', 'db_notap.py', 'db_nopresum.py')), [])}
