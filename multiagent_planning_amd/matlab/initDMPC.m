function [p,v,a] = initDMPC(po,pf,h,k_hor,K)
% Shadows dmpc/matlab/initDMPC.m (same signature): straight-line prediction p(:,i) = po + t_i (pf-po)/10, v = a = 0.
prm = dmpc_params_struct(0, h, k_hor, 0.35, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);     % context only
[p,v,a] = dmpc_mex('init_batch', prm, po(:), pf(:));
end
