import numpy as np, os, sys
from tests import util, refdrive as rd, cases
from roms_amd import hiplib
app, cs = rd.make_case("overflow_small"); cs["hadv"], cs["vadv"] = ("U3", "U3"), ("C4", "C4")
saved = rd.quiet(); R = rd.reference(app, cs); rd.unquiet(saved)
O = rd.oracle_from(R, cs); O.start(); O.main3d_step(2)
b = R.bounds(0); nd = cs["ndtfast"]
w = np.stack([R.table(5, 2 * nd), R.table(6, 2 * nd)])
cfg = cases.hip_cfg(cs, R.table(7, 8)[0], b[58], w, R.table(1, cs["N"]), R.table(2, cs["N"]), R.table(3, cs["N"]+1), R.table(4, cs["N"]+1))
H = hiplib.Context(cfg, util.EMU_LIB)
s = O.step; s.nstp = 1 + (s.iic - 1) % 2; s.nnew = 3 - s.nstp; s.nrhs = s.nstp
util.push_state(O, H)
for kname in ["set_massflux", "rho_eos", "set_vbc", "omega", "set_zeta"]:
    O.call(kname); H.call(kname)
N=cs["N"]; ni=O.ni; nj=O.nj
T = O.field("t").copy().reshape(2,3,N,nj,ni); Hu=O.field("Huon").reshape(N,nj,ni); Hv=O.field("Hvom").reshape(N,nj,ni)
Hz=O.field("Hz").reshape(N,nj,ni); W=O.field("W").reshape(N+1,nj,ni); pm=O.field("pm").reshape(nj,ni); pn=O.field("pn").reshape(nj,ni)
O.call("pre_step3d"); H.call("pre_step3d")
A=H.download("t").reshape(2,3,N,nj,ni); B=O.field("t").reshape(2,3,N,nj,ni)
idx=np.argwhere(A!=B); it,n,k,j,i = [int(x) for x in idx[2]]     # an interior column
print("RESULT point", (it,n,k,j,i), "emu %.17e orc %.17e" % (A[it,n,k,j,i], B[it,n,k,j,i]))
t = T[it, s.nstp-1]; tn = T[it, s.nnew-1]
Gamma=1.0/6.0; dt=cs["dt"]; cff=(1-Gamma)*dt; c1=0.5+Gamma; c2=0.5-Gamma
def fx(ii, jj):   # U3 flux at u-point ii (closed west/east: Istr=1, Iend=Lm)
    Lm=cs["Lm"]
    def grad(m):
        if m == 0: m = 1
        if m == Lm+2: m = Lm+1
        return t[k,jj,m]-t[k,jj,m-1]
    cm=grad(ii)-grad(ii-1); c0=grad(ii+1)-grad(ii); h=Hu[k,jj,ii]
    return h*0.5*(t[k,jj,ii-1]+t[k,jj,ii]) - (1.0/6.0)*(cm*max(h,0.0)+c0*min(h,0.0))
def fe(ii, jj):
    Mm=cs["Mm"]
    def grad(m):
        if m == 0: m = 1
        if m == Mm+2: m = Mm+1
        return t[k,m,ii]-t[k,m-1,ii]
    cm=grad(jj)-grad(jj-1); c0=grad(jj+1)-grad(jj); h=Hv[k,jj,ii]
    return h*0.5*(t[k,jj-1,ii]+t[k,jj,ii]) - (1.0/6.0)*(cm*max(h,0.0)+c0*min(h,0.0))
FX0,FXp,FE0,FEp = fx(i,j),fx(i+1,j),fe(i,j),fe(i,j+1)
t3h = Hz[k,j,i]*(c1*t[k,j,i]+c2*tn[k,j,i]) - cff*pm[j,i]*pn[j,i]*(FXp-FX0+FEp-FE0)
print("RESULT fluxes", FX0, FXp, FE0, FEp, "t3h", t3h)
def fc(kk):   # C4 vertical flux at w-level kk (1-based interface index: between rho levels kk and kk+1), arrays 0-based
    if kk<=0 or kk>=N: return 0.0
    Tc=lambda q: t[q-1,j,i]; Wk=W[kk,j,i]
    if kk==1: return Wk*(0.5*Tc(1)+7/12*Tc(2)-1/12*Tc(3))
    if kk==N-1: return Wk*(0.5*Tc(N)+7/12*Tc(N-1)-1/12*Tc(N-2))
    return Wk*(7/12*(Tc(kk)+Tc(kk+1))-1/12*(Tc(kk-1)+Tc(kk+2)))
kk=k+1
DC=1.0/(Hz[k,j,i]-cff*pm[j,i]*pn[j,i]*(Hu[k,j,i+1]-Hu[k,j,i]+Hv[k,j+1,i]-Hv[k,j,i]+(W[kk,j,i]-W[kk-1,j,i])))
t3=DC*(t3h-cff*pm[j,i]*pn[j,i]*(fc(kk)-fc(kk-1)))
print("RESULT numpy t3 %.17e" % t3, "Huon", Hu[k,j,i], Hu[k,j,i+1], "W", W[kk,j,i])
from fractions import Fraction as Fr
def fma(a,b,c): return float(Fr(a)*Fr(b)+Fr(c))
print("RESULT emu   %.17e" % A[it,n,k,j,i])
div2=(FXp-FX0)+(FEp-FE0)
v=DC*((Hz[k,j,i]*(c1*t[k,j,i]+c2*tn[k,j,i]) - cff*pm[j,i]*pn[j,i]*div2)-cff*pm[j,i]*pn[j,i]*(fc(kk)-fc(kk-1)))
print("RESULT pair-sum %.17e" % v)
pmn=pm[j,i]*pn[j,i]
v=DC*(t3h-(cff*pmn)*(fc(kk)-fc(kk-1))); print("RESULT cfv1 %.17e"%v)
v=(t3h-(cff*pmn)*(fc(kk)-fc(kk-1)))/(1.0/DC); print("RESULT div %.17e"%v)
print("RESULT tn, t", tn[k,j,i], t[k,j,i], "Hv", Hv[k,j,i], Hv[k,j+1,i], "c2*tn", c2*tn[k,j,i])
print("RESULT t col", t[:,j,i])
