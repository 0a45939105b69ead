"""One-off source transformation used when the HIP kernels were made generic over the model's float type."""
import re, sys
def convert(path, extra=None):
    s=open(path).read()
    s=re.sub(r'(?<![\w.])(\d+\.\d*(?:[eE][+-]?\d+)?|\d+(?:[eE][+-]?\d+)|\.\d+(?:[eE][+-]?\d+)?)f\b', r'real(\1)', s)
    s=re.sub(r'\bfloat4\b','real4',s)
    s=re.sub(r'\bfloat2\b','real2',s)
    s=re.sub(r'\bfloat\b','real',s)
    s=s.replace('fabsf(','rabs(').replace('fminf(','rmin(').replace('fmaxf(','rmax(').replace('tanhf(','rtanh(')
    if extra: s=extra(s)
    open(path,'w').write(s)
for p in sys.argv[1:]:
    convert(p)
