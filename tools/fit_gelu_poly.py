"""Minimax (LP) fit of the packed-fp32 polynomial GELU of csam_common.h: Phi(x) ~ 0.5 + xc*R(xc^2), xc = clamp(x, +-c),
constrained to Phi(c) = 1 so that gelu(x) = x exactly beyond the clamp.  Prints the fp32-Horner error of each (c, degree)."""
import numpy as np
from scipy.special import erf
from scipy.optimize import linprog
from numpy.polynomial import chebyshev as C
def phi(x): return 0.5*(1+erf(x/np.sqrt(2)))
def fit(c,deg,n=3000):
    k=np.arange(n); t=np.cos(np.pi*(k+0.5)/n)
    u=(t+1)/2*c*c; x=np.sqrt(u)
    target=(phi(x)-0.5)/x
    V=C.chebvander(t,deg)           # R(u) = V@coef
    # error in gelu for |x|<=c : x * x*(R - target)  (gelu = x*Phi, Phi = .5 + x R)  -> weight x^2 ; we bound Phi error*max(|x|,1)?
    wgt=x*np.maximum(x,0.3)
    A=np.zeros((2*n,deg+2)); b=np.zeros(2*n)
    A[:n,:deg+1]=V*wgt[:,None]; A[:n,-1]=-1; b[:n]=target*wgt
    A[n:,:deg+1]=-V*wgt[:,None]; A[n:,-1]=-1; b[n:]=-target*wgt
    Aeq=np.zeros((1,deg+2)); Aeq[0,:deg+1]=C.chebvander(np.array([1.0]),deg)[0]; beq=[0.5/c]
    cost=np.zeros(deg+2); cost[-1]=1
    res=linprog(cost,A_ub=A,b_ub=b,A_eq=Aeq,b_eq=beq,bounds=[(None,None)]*(deg+2),method="highs")
    coef=res.x[:deg+1]
    pc=C.cheb2poly(coef)
    base=np.array([-1.0, 2.0/(c*c)]); powt=np.array([1.0]); pu=np.zeros(deg+1)
    for i,a in enumerate(pc):
        pu[:len(powt)]+=a*powt; powt=np.convolve(powt,base)
    return pu,res.x[-1]
def evalerr(pu,c):
    xs=np.linspace(-10,10,800001).astype(np.float32)
    xc=np.clip(xs,-np.float32(c),np.float32(c))
    uu=(xc*xc).astype(np.float32)
    r=np.full_like(uu,np.float32(pu[-1]))
    for a in pu[-2::-1]:
        r=(r*uu+np.float32(a)).astype(np.float32)
    ph=(xc*r+np.float32(0.5)).astype(np.float32)
    g=(xs*ph).astype(np.float32)
    ref=xs.astype(np.float64)*phi(xs.astype(np.float64))
    err=np.abs(g-ref)
    return err.max(), xs[err.argmax()], np.abs(err[np.abs(xs)<3]).max()
for c in (3.6,3.8,4.0,4.2,4.4,4.6):
    for deg in (6,7,8,9):
        pu,tt=fit(c,deg)
        e,xa,e3=evalerr(pu,c)
        print(f"c={c} deg={deg}: lp bound {tt:.2e}; f32 max|gelu err|={e:.2e} at x={xa:.2f}; |x|<3: {e3:.2e}")
        if (c,deg) in [(4.0,7),(4.2,8),(4.4,9),(3.8,7),(4.0,8)]:
            print("   coef:", ", ".join("%.9ef"%v for v in pu))
