# timing experiment (WRONG RESULTS on purpose): db_core without the strip barriers and without re-staging: the bare MFMA stream of the ConvT3 contraction
import runpy, os
_d = os.path.dirname(os.path.abspath(__file__))
PATCH = {'decoder.hip': sum((runpy.run_path(os.path.join(_d, f))['PATCH']['decoder.hip'] for f in ('db_nobarrier.py', 'db_nostage.py', 'db_nogather.py', 'db_notap.py', 'db_nopresum.py')), [])}
