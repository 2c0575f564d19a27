function [p,v,a,success,outbound] = solveEllipDMPC(po,pf,vo,ao,n,h,l,K,rmin,pmin,pmax,alim,A,A_initp,Delta,Q1,S1,E1,E2,order)
% Drop-in replacement of dmpc/matlab/solveEllipDMPC.m over dmpc_mex.
prm = dmpc_params_struct(5, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, -5e4);
[p,v,a,st] = dmpc_mex('solve_one', prm, l, n, po, vo, ao, pf);
success = double(bitand(st,1) ~= 0); outbound = 0;
if ~success, p = []; v = []; a = []; end
end
