function [pk,vk,ak,K_T_used,feasible,reached] = dmpc_transition(prm, po, pf, K_T_max, error_tol)
% The whole MPC loop `for k = 1:K_T` of dmpc/matlab/dmpc_soft_bound.m:115-148 (test/failure_rate.m:99-127) for one trial on the
% GPU: initDMPC at k = 1, one batched solve of all agents per step, l = new_l, ReachedGoal.  po, pf: 1 x 3 x N as the
% reference holds them; pk, vk, ak: 3 x K_T_max x N (columns beyond K_T_used are zero).
N = size(po,3);
[pk,vk,ak,K_T_used,st] = dmpc_mex('transition', prm, reshape(po,3,N), reshape(pf,3,N), K_T_max, error_tol);
feasible = bitand(st, 2+4+8+16+32) == 0;
reached = bitand(st, 256) ~= 0;
end
