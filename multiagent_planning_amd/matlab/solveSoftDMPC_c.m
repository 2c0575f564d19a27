function [p,v,a,success] = solveSoftDMPC_c(po,pf,vo,ao,n,h,l,K,rmin,pmin,pmax,alim,A,A_initp,Delta,Q1,S1,E1,E2,order)
% Drop-in replacement of dmpc/matlab/solveSoftDMPC_c.m over dmpc_mex (slack penalties -1e4 (K/k)^2, 1e6 (K/k)^2 inside the library).
prm = dmpc_params_struct(11, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, -1e4);
[p,v,a,st] = dmpc_mex('solve_one', prm, l, n, po, vo, ao, pf);
success = double(bitand(st,1) ~= 0);
if ~success, p = []; v = []; a = []; end   % solveSoftDMPC_c.m:80-87
end
