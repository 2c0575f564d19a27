function [p,v,a,success,outbound] = solveSoftDMPC(po,pf,vo,ao,n,h,l,K,rmin,pmin,pmax,alim,A,A_initp,Delta,Q1,S1,E1,E2,order)
% Drop-in replacement of dmpc/matlab/solveSoftDMPC.m over dmpc_mex.
prm = dmpc_params_struct(6, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, -1e5);
[p,v,a,st,info] = dmpc_mex('solve_one', prm, l, n, po, vo, ao, pf);
success = double(bitand(st,1) ~= 0); outbound = 0;
if ~success, p = []; v = []; a = []; outbound = double(info(1) == 0); end   % solveSoftDMPC.m:88-96
end
