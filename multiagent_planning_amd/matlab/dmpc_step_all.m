function [new_l,pk1,vk1,ak1,status] = dmpc_step_all(prm, l, pk_prev, vk_prev, ak_prev, pf)
% Batched replacement of the agent loop `for n = 1:N` of dmpc/matlab/dmpc_soft_bound.m:116-135:
% one GPU launch solves every agent of the MPC step.  pk_prev etc. are 3 x N, pf is 1 x 3 x N.
N = size(l,3);
[new_l,V,A,status] = dmpc_mex('step_batch', prm, l, pk_prev, vk_prev, ak_prev, reshape(pf,3,N));
pk1 = squeeze(new_l(:,1,:)); vk1 = squeeze(V(:,1,:)); ak1 = squeeze(A(:,1,:));
end
